#!/bin/bash
# VALU / SALU instructions per wave of the fused kernels: tools/r6_pmc_front.sh <tag> [workload]
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmcf_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs --parity-scans 0 --workload ${2:-cfg3} --front 1 > $OUT/log 2>&1
cd $REPO; python tools/pmc_summary.py $OUT | grep -A12 "## k_front\|## k_label_front" | grep "##\|per wave\|WAIT_ANY\|WAVE_CYCLES\|ACTIVE_INST_VALU"
