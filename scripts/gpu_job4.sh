#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
python bench.py --workload cant --no-config5 --no-cpu-baseline 2>/dev/null | tail -n 1
