#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4o
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r4o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4o/pytest.log
timeout 900 python tools/fuzz_more.py 500000 560000 > gpurun_out/r4o/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r4o/fuzz.log
python - <<'PY' > gpurun_out/r4o/adapter.json 2>&1
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import urban_road_filter_amd as u, bench
print(json.dumps(bench.adapter_e2e(u)))
PY
cat gpurun_out/r4o/adapter.json
