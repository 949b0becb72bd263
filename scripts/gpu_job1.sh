#!/bin/bash
# first GPU job of round 2: parity suite + per-class times + bench lines on the re-fitted stand-ins
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/j1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/j1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/j1/pytest.log
for w in scircuit mac_econ cant webbase; do
  timeout 300 python scripts/class_times.py $w > gpurun_out/j1/class_$w.log 2>&1
  timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/j1/bench_$w.log 2>&1
done
tail -n 3 gpurun_out/j1/pytest.log
cat gpurun_out/j1/class_*.log
