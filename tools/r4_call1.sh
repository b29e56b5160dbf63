#!/bin/bash
# round 4, first GPU call: VALU issue microbenchmark, the whole GPU suite, the default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4a
timeout 120 tools/bench_micro/valubench > gpurun_out/r4a/valubench.txt 2>&1; echo "valubench rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x --maxfail=1 > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r4a/pytest.log
timeout 600 python bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r4a/bench.json; tail -5 gpurun_out/r4a/bench.err
