#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 4
bash scripts/gpu_ab.sh "base" "base"
bash scripts/gpu_job4.sh 2>&1 | grep -v amdgpu | tail -n 4
