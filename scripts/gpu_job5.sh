#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
make -s clean >/dev/null 2>&1
make -s PHASE_CLOCKS=1 speck_amd/libspeck_amd.so 2>&1 | grep -v warning | head
for w in webbase scircuit; do python scripts/analysis_clocks.py $w 2>&1 | grep -v amdgpu.ids; done
