#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/calib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/fetch_calib.hip -o gpurun_out/calib/fetch_calib
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '+')
  rm -rf gpurun_out/_c
  rocprofv3 --kernel-trace --pmc $pass -d gpurun_out/_c -o r -- gpurun_out/calib/fetch_calib > gpurun_out/calib/$tag.log 2>&1
  python scripts/rocpd_pmc.py $(find gpurun_out/_c -name "*.db" | head -n 1) gpurun_out/calib/$tag.csv > /dev/null || tail -n 3 gpurun_out/calib/$tag.log
done
rm -rf gpurun_out/_c gpurun_out/calib/fetch_calib
cat gpurun_out/calib/*.csv | cut -c1-160
timeout 300 python bench.py --workload uniform --no-config5 --no-cpu-baseline | cut -c1-700
