#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5
bash scripts/gpu_ab.sh "base" "fork_min_us=100000" "fork_min_us=0" "base"
