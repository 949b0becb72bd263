#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/pmc_passes.sh me_plain mac_econ --opt flat_tiny=0
bash scripts/pmc_passes.sh me_piped mac_econ --opt flat_tiny=2
grep -h "num_tiny" gpurun_out/pmc/me_*_pass*.csv | cut -c1-300
