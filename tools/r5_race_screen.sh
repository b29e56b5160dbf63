#!/bin/bash
# r5 race screen: the GPU parity suites on builds whose occupancy differs from the shipped one (k_ring at 4 / 5 waves per SIMD,
# k_split at 4 / 8): a label that depends on occupancy is what a missing barrier looks like
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5_race
for name in ring4 ring5 split4 split8; do
  URF_LIB_PATH="$GRAFT_REPO_ROOT/tools/ab/liburf_hip_$name.so" timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batches.py tests/test_gpu_fuzz.py tests/test_gpu_async.py -m gpu -q --maxfail=50 > gpurun_out/r5_race/$name.log 2>&1
  echo "$name rc=$? $(tail -1 gpurun_out/r5_race/$name.log)"
done
