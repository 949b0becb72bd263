#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
make -s clean >/dev/null 2>&1
make -s PHASE_CLOCKS=1 speck_amd/libspeck_amd.so 2>&1 | grep -v warning | head
for w in mac_econ scircuit; do python scripts/phase_clocks.py $w flat_tiny=0 concurrent_classes=0; done
