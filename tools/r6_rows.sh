#!/bin/bash
# row-major organised sweeps (bench.py --workload ring_major): bench line + rocprofv3 kernel trace.  tools/r6_rows.sh <tag>
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/rows_$1; mkdir -p $OUT
cd $REPO; python bench.py --workload ring_major --steps 20 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs --workload ring_major > $OUT/trace.log 2>&1
cd $REPO; python tools/rocprof_summary.py $(find $OUT -name "*results.db" | head -1) > gpurun_out/rows_$1_kernel_stats.txt 2>&1; head -24 gpurun_out/rows_$1_kernel_stats.txt
python -c "
import json;b=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1]);print(b['value'],b['ms_per_step'],b['config'].get('front_scans_per_gpu'),b.get('kernel_ms'))"
