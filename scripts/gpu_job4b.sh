#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
bash scripts/gpu_ab.sh "base" "base"
for w in scircuit mac_econ cant webbase; do python scripts/class_times.py $w 2>/dev/null | head -n 1; done
