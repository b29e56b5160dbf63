#!/bin/bash
# One GPU session that produces everything profiles/ needs for a round:
#   <out>/bench.json           un-profiled bench line (with cpu_baseline)
#   <out>/trace/*_results.db   rocprofv3 --kernel-trace --stats of the same command
#   <out>/pmc/...              PMC passes (tools/pmc_profile.sh)
#   tools/profile_round.sh gpurun_out/r1_final
set -u
OUT=${1:-gpurun_out/round}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$REPO/$OUT";; esac
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python bench.py --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 400 "$OUT/bench.json"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-other-configs > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
bash "$REPO/tools/pmc_profile.sh" "$OUT/pmc" | grep pass
