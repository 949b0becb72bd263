#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 4
