#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
mkdir -p gpurun_out/stress
timeout 900 python tests/tools/stress_gpu.py 150 1 > gpurun_out/stress/default.log 2>&1; echo "default rc=$?"; tail -n 1 gpurun_out/stress/default.log
timeout 900 python tests/tools/stress_gpu.py 120 2 nf_min_ops=1 xcd_aware=7 fork_min_us=0 > gpurun_out/stress/nf1.log 2>&1; echo "nf1 rc=$?"; tail -n 1 gpurun_out/stress/nf1.log
timeout 900 python tests/tools/stress_gpu.py 100 3 num_global_passes=100000000 sym_bitmap_ratio=1000000 > gpurun_out/stress/windows.log 2>&1; echo "windows rc=$?"; tail -n 1 gpurun_out/stress/windows.log
grep -h BAD gpurun_out/stress/*.log | head
