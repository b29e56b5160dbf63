#!/bin/bash
# r5: LDS padding A/B (k_split staging slots, sort bucket counters), k_ring at 7 waves, then the GPU suite and one PMC pass
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5e
AB_ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-other-configs --no-outputs" bash tools/ab_bench.sh gpurun_out/r5e/ab tools/ab/liburf_hip_pad64.so tools/ab/liburf_hip_cntnopad.so tools/ab/liburf_hip_ringw7.so tools/ab/liburf_hip_new.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e/ab.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -x > gpurun_out/r5e/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5e/pytest.log
bash tools/pmc_quick.sh gpurun_out/r5e/pmc cfg3 2>&1 | tail -3
python tools/pmc_summary.py gpurun_out/r5e/pmc > gpurun_out/r5e/pmc.txt 2>&1
grep -A16 "^## k_split\b\|^## k_star_sort_small\|^## k_ring\b" gpurun_out/r5e/pmc.txt | grep "##\|BANK\|IDX_ACTIVE\|WAIT_ANY\|WAVE_CYCLES\|per wave"
