#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
mkdir -p gpurun_out/j2
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in scircuit mac_econ cant webbase; do
  timeout 300 python scripts/class_times.py $w > gpurun_out/j2/class_$w.log 2>&1
  rm -rf gpurun_out/_p_$w
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/_p_$w/trace -o r -- python bench.py --workload $w --no-cpu-baseline > gpurun_out/j2/bench_${w}_rocprof.log 2>&1
  python scripts/rocpd_summary.py $(find gpurun_out/_p_$w/trace -name "*.db" | head -n 1) gpurun_out/j2/${w}_kernel_stats.csv > /dev/null
  rm -rf gpurun_out/_p_$w
done
cat gpurun_out/j2/class_*.log
