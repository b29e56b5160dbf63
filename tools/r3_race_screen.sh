#!/bin/bash
# Race screen (VERDICT r2 #5c): the GPU parity tests on builds whose occupancy differs from the shipped one
# (a label that depends on occupancy is what a missing barrier looks like), then the full default bench.
#   tools/r3_race_screen.sh <out_dir> <lib.so>...
set -u
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"; mkdir -p "$OUT"
for lib in "$@"; do
  name=$(basename "$lib" .so)
  URF_LIB_PATH="$REPO/$lib" timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batches.py tests/test_gpu_fuzz.py tests/test_gpu_async.py -m gpu -q --maxfail=50 > "$OUT/$name.log" 2>&1
  echo "$name rc=$? $(tail -1 "$OUT/$name.log")"
done
timeout 900 python bench.py --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], d["kernel_ms"])
print("cpu", d["cpu_baseline"])
for k in d:
    if k.startswith("e2e") or k == "outputs_ms_per_batch": print(k, d[k])
for k, v in d.get("other_configs", {}).items(): print(k, {a: v[a] for a in ("scans_per_s", "ms_per_step", "frac")}, v["kernel_ms"])
PY
