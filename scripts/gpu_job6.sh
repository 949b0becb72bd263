#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/pmc_passes.sh wb webbase --opt concurrent_classes=0 > /dev/null 2>&1
bash scripts/pmc_passes.sh ct cant > /dev/null 2>&1
for t in wb ct; do for i in 1 2 3; do cut -c1-260 gpurun_out/pmc/${t}_pass$i.csv | grep -v "copyBuffer\|fillBuffer\|done_kernel\|sym_bitmap_kernel\|Block<1024>"; done; done
