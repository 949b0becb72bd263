#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --workload webbase --no-cpu-baseline --no-config5 --steps 20 > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
python scripts/rocpd_summary.py $db 2>/dev/null | head -30
