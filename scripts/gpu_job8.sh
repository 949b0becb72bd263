#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
bash scripts/collect_profiles.sh r02 "scircuit mac_econ cant webbase" > gpurun_out/collect.log 2>&1
tail -n 25 gpurun_out/collect.log
python bench.py > gpurun_out/profiles/r02_bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/profiles/r02_bench_default.json
python scripts/multiwindow_time.py _old_tmp 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles/r02_multiwindow_before_after.txt
python scripts/multiwindow_time.py . 2>&1 | grep -v amdgpu.ids >> gpurun_out/profiles/r02_multiwindow_before_after.txt
