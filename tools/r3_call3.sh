cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
URF_LIB_PATH=$PWD/tools/ab/liburf_hip_t1024.so timeout 600 python -m pytest tests -m gpu -q --maxfail=15 --deselect tests/test_abi.py > gpurun_out/r3c/tests_t1024.log 2>&1; echo "t1024 pytest rc=$?"; tail -5 gpurun_out/r3c/tests_t1024.log
bash tools/r3_call.sh gpurun_out/r3c - urban_road_filter_amd/liburf_hip.so tools/ab/liburf_hip_t1024.so
