#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python scripts/multiwindow_time.py _old_tmp 2>&1 | grep -v amdgpu.ids
python scripts/multiwindow_time.py . 2>&1 | grep -v amdgpu.ids
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5
