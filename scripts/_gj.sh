#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
WL="cant scircuit" bash scripts/gpu_ab_libs.sh base nfwin base nfwin
