#!/bin/bash
# A/B bench of several builds of the library inside ONE gpurun call (box-to-box spread is larger
# than most kernel changes): tools/ab_bench.sh <out_dir> <lib.so>... ; prints kernel_ms per library.
set -u
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$REPO/$OUT";; esac
mkdir -p "$OUT"
cd "$REPO"
EXTRA=${AB_ARGS:-"--steps 8 --warmup 2 --no-cpu-baseline"}
for round in 1 2; do
for lib in "$@"; do
  name=$(basename "$lib" .so)
  URF_LIB_PATH="$REPO/$lib" timeout 300 python bench.py $EXTRA > "$OUT/$name.$round.json" 2> "$OUT/$name.$round.err" || echo "FAILED $name: $(tail -2 $OUT/$name.$round.err)"
  python - "$OUT/$name.$round.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s %8.4f ms  %s" % (sys.argv[2], d["ms_per_step"], " ".join("%s=%.3f" % (k[2:], v) for k, v in d["kernel_ms"].items())))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
done
