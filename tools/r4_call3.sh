#!/bin/bash
# round 4: phase clocks of the big kernels (cfg3) + A/B of the ring table's wave skip
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4i
URF_LIB_PATH=$PWD/tools/ab/liburf_hip_clk.so timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-outputs --no-other-configs --parity-scans 0 > gpurun_out/r4i/clk.log 2>&1
grep -E "^k_|phase|sector" gpurun_out/r4i/clk.log | sort | uniq -c | sort -k2 | head -150 > gpurun_out/r4i/clk_summary.txt
wc -l gpurun_out/r4i/clk.log
AB_WORKLOADS="default_roi" bash tools/r3_call.sh gpurun_out/r4i - tools/ab/liburf_hip_org1.so tools/ab/liburf_hip_tabskip.so
