#!/bin/bash
# round 4: extended fuzz at HEAD + the whole GPU suite + the default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4m
timeout 900 python tools/fuzz_more.py 400000 415000 > gpurun_out/r4m/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r4m/fuzz.log
timeout 600 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r4m/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4m/pytest.log
timeout 600 python bench.py > gpurun_out/r4m/bench.json 2> gpurun_out/r4m/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r4m/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('valu_issue'), d['adapter_e2e_ms'])"
