#!/bin/bash
# round 5 evidence at HEAD: smoke, cfg3 bench + kernel trace + PMC passes (tools/profile_round.sh), the GPU suite
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5_final
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/profile_round.sh gpurun_out/r5_final 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r5_final/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5_final/pytest.log
