#!/bin/bash
# round 4, final evidence at HEAD: profile round (cfg3 bench + trace + PMC; cfg5 trace + PMC), the GPU suite, a long fuzz
cd "$GRAFT_REPO_ROOT"
bash tools/r4_profile.sh 2>&1 | grep -v amdgpu.ids
mkdir -p gpurun_out/r4_final
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r4_final/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4_final/pytest.log
timeout 1200 python tools/fuzz_more.py 600000 800000 > gpurun_out/r4_final/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r4_final/fuzz.log
