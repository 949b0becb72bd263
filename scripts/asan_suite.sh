#!/bin/bash
# The host side of the library under AddressSanitizer + UndefinedBehaviorSanitizer (make ASAN=1):
#   bash scripts/asan_suite.sh cpu   -- the CPU suite (-m "not gpu") + tests/cpp/test_gather_layout.cpp, here or anywhere
#   bash scripts/asan_suite.sh gpu   -- a slice of the GPU suite through the instrumented host code (GPU box)
# Output: gpurun_out/asan/<mode>.log; the script fails if a sanitizer reported anything.
MODE=${1:-cpu}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONPATH=$PWD
mkdir -p gpurun_out/asan
if [ "$MODE" = cpu ]; then
  make ASAN=1 -j8 speck_amd/libspeck_amd_asan.so > gpurun_out/asan/build.log 2>&1 || { echo "ASAN build failed"; tail gpurun_out/asan/build.log; exit 2; }
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
  export SPECK_LIB=$PWD/speck_amd/libspeck_amd_asan.so
else
  # (beside a GPU: UBSan alone -- ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and aborts the process
  #  unless the DEVICE code is an xnack+ sanitizer build as well)
  make UBSAN=1 -j8 speck_amd/libspeck_amd_ubsan.so > gpurun_out/asan/build.log 2>&1 || { echo "UBSAN build failed"; tail gpurun_out/asan/build.log; exit 2; }
  RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
  export SPECK_LIB=$PWD/speck_amd/libspeck_amd_ubsan.so
fi
# (python itself leaks by design; the HIP runtime maps device memory ASan cannot shadow)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:log_path=$PWD/gpurun_out/asan/asan_$MODE
export UBSAN_OPTIONS=print_stacktrace=1:log_path=$PWD/gpurun_out/asan/ubsan_$MODE
rm -f gpurun_out/asan/asan_$MODE.* gpurun_out/asan/ubsan_$MODE.*
LOG=gpurun_out/asan/$MODE.log
if [ "$MODE" = cpu ]; then
  # (no device for this process even on a GPU box: ROCm's ASan runtime aborts as soon as the HIP runtime allocates)
  HIP_VISIBLE_DEVICES=-1 ROCR_VISIBLE_DEVICES=-1 LD_PRELOAD=$RT timeout 1500 python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider > $LOG 2>&1
  rc=$?
  /opt/rocm/bin/hipcc -std=c++17 -O1 -g -fsanitize=address,undefined -Ispeck_amd/csrc tests/cpp/test_gather_layout.cpp -o gpurun_out/asan/test_gather_layout >> $LOG 2>&1 \
    && gpurun_out/asan/test_gather_layout >> $LOG 2>&1
  rc2=$?
else
  # (single-process tests only: the launcher tests start ranks of their own, which would inherit the preload)
  LD_PRELOAD=$RT timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_walk.py -x -q -m gpu -p no:cacheprovider --timeout 200 \
      -k "tiny or early or matout or reuse_rules or randomised or survives or unsorted or row_view or walk_call or freed" > $LOG 2>&1
  rc=$?; rc2=0
fi
tail -n 3 $LOG
found=$(ls gpurun_out/asan/asan_$MODE.* gpurun_out/asan/ubsan_$MODE.* 2>/dev/null | wc -l)
echo "pytest rc=$rc cpp rc=$rc2 sanitizer reports: $found"
[ "$found" != 0 ] && head -n 60 $(ls gpurun_out/asan/asan_$MODE.* gpurun_out/asan/ubsan_$MODE.* | head -n 3)
[ $rc = 0 ] && [ $rc2 = 0 ] && [ "$found" = 0 ]
