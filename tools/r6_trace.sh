#!/bin/bash
# rocprofv3 kernel trace of one workload / front mode: tools/r6_trace.sh <tag> <workload> <front>
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/trace_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs --workload $2 --front $3 > $OUT/trace.log 2>&1
cd $REPO; python tools/rocprof_summary.py $(find $OUT -name "*results.db" | head -1) > gpurun_out/$1_kernel_stats.txt 2>&1; head -30 gpurun_out/$1_kernel_stats.txt
