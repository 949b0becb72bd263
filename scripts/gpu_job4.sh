#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5
WL="scircuit mac_econ" bash scripts/gpu_ab.sh "spin_wait=1" "spin_wait=0" "spin_wait=1" "spin_wait=0"
python scripts/latency_floor.py
