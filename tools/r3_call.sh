#!/bin/bash
# One gpurun call of round 3: GPU tests on the in-tree library, then A/B benches of several builds.
#   tools/r3_call.sh <out_dir> <pytest args or "-"> <lib.so>...
set -u
OUT=$1; shift
PYT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
mkdir -p "$OUT"
if [ "$PYT" != "-" ]; then
  timeout 900 python -m pytest tests -m gpu -q --maxfail=15 $PYT > "$OUT/tests.log" 2>&1
  echo "pytest rc=$?"; tail -15 "$OUT/tests.log"
fi
for w in ${AB_WORKLOADS:-cfg3 default_roi}; do
  echo "== $w"
  AB_ARGS="--workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs" bash tools/ab_bench.sh "$OUT/ab_$w" "$@"
done
