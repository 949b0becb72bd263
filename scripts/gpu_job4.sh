#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5
bash scripts/gpu_ab.sh "base"
for w in scircuit mac_econ cant webbase; do python scripts/class_times.py $w 2>/dev/null | head -n 1; done
