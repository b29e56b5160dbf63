#!/bin/bash
# A/B of the callback path inside ONE gpurun call: per-kernel times of a single resident sweep (bench.py --workload cfg2)
# and sweeps/s / latency of the native submit / collect loop (tools/host_times.py) per library.
#   tools/ab_callback.sh <lib.so>...
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
for round in 1 2; do
for lib in "$@"; do
  name=$(basename "$lib" .so)
  echo "== $name (round $round)"
  URF_LIB_PATH="$REPO/$lib" timeout 120 python bench.py --workload cfg2 --steps 50 --warmup 5 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs 2>/dev/null | python -c '
import json, sys
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print("  resident sweep %.4f ms  %s" % (d["ms_per_step"], " ".join("%s=%.4f" % (k[2:], v) for k, v in d["kernel_ms"].items())))
'
  [ -n "${AB_NO_STREAM:-}" ] || URF_LIB_PATH="$REPO/$lib" timeout 60 python tools/host_times.py 600 2>&1 | grep -v "^host" | grep -E "in flight (4|1)" | sed "s/^/  /"
done
done
