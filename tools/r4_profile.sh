#!/bin/bash
# round 4 evidence: cfg3 bench + kernel trace + PMC passes (tools/profile_round.sh), then the same trace / PMC for cfg5
cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh gpurun_out/r4_final
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4_final"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace_cfg5" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-other-configs --no-outputs > "$OUT/trace_cfg5.log" 2>&1
echo "cfg5 trace rc=$?"
bash "$GRAFT_REPO_ROOT/tools/pmc_profile.sh" "$OUT/pmc_cfg5" --workload cfg5 | grep pass
