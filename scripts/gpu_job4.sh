#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_ab.sh "xcd_aware=0" "xcd_aware=2" "xcd_aware=1" "xcd_aware=0" "xcd_aware=2"
