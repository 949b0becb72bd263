#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5
bash scripts/gpu_ab.sh "base" "base"
for w in scircuit mac_econ; do python scripts/class_times.py $w | head -n 2; done
