#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash scripts/gpu_ab_libs.sh base bm1half base bm1half
