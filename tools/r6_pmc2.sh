#!/bin/bash
# extra counters of k_front for one workload: tools/r6_pmc2.sh <tag> <workload> <counters...>
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; wl=$2; shift 2
OUT=$REPO/gpurun_out/pmc2_$tag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs --parity-scans 0 --workload $wl --front 1 > $OUT/log 2>&1
cd $REPO; python tools/pmc_summary.py $OUT | grep -A14 "## k_front$" | grep -v "per wave"
