#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5
bash scripts/gpu_ab.sh "base" "base"
bash scripts/pmc_passes.sh me_rank2 mac_econ > /dev/null 2>&1
grep -h "num_tiny" gpurun_out/pmc/me_rank2_pass[12].csv | cut -c1-200
