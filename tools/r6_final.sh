#!/bin/bash
# round 6 evidence at HEAD: smoke, cfg3 bench + kernel trace + PMC passes (tools/profile_round.sh), the sensor-like workload's trace
cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/r6_final}
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/profile_round.sh $OUT 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace_sensor -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload sensor --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-other-configs > $GRAFT_REPO_ROOT/$OUT/trace_sensor.log 2>&1
echo "sensor trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace_rows -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload ring_major --steps 5 --warmup 3 --no-cpu-baseline --no-e2e --no-other-configs > $GRAFT_REPO_ROOT/$OUT/trace_rows.log 2>&1
echo "row-major trace rc=$?"
