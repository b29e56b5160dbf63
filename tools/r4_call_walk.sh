#!/bin/bash
# round 4, the walk rework: GPU suite on the in-tree build, 20 000 fuzz clouds (single sweeps: k_star_walk_few), default bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4walk
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r4walk/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4walk/pytest.log
timeout 600 python tools/fuzz_more.py 900000 920000 > gpurun_out/r4walk/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r4walk/fuzz.log
timeout 900 python bench.py > gpurun_out/r4walk/bench.json 2> gpurun_out/r4walk/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4walk/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['kernel_ms'])
print({k:v for k,v in d.items() if 'e2e' in k and not isinstance(v,dict)})
print({k:(v['ms_per_step'], v['kernel_ms']) for k,v in d['other_configs'].items()})
PY
