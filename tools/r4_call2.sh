#!/bin/bash
# round 4, second GPU call: VALU microbenchmark (more encodings), the whole GPU suite, the adapter's timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4b
timeout 200 tools/bench_micro/valubench > gpurun_out/r4b/valubench.txt 2>&1; echo "valubench rc=$?"
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r4b/pytest.log
python - <<'PY' > gpurun_out/r4b/adapter.json 2>&1
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import urban_road_filter_amd as u, bench
print(json.dumps(bench.adapter_e2e(u)))
PY
cat gpurun_out/r4b/adapter.json
