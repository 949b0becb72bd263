#!/bin/bash
# kernel stats of scripts/dev/ab.py for one workload and option set: bash scripts/dev/trace_ab.sh nlpkkt:0.2 "one_walk_hash=1"
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
rm -rf gpurun_out/_tr; rocprofv3 --kernel-trace --stats -d gpurun_out/_tr -o r -f csv -- python scripts/dev/ab.py "$1" "$2" 2>&1 | grep -v "^[WE]2026\|amdgpu.ids" | tail -2
f=$(find gpurun_out/_tr -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:8]:
    print("%-70s calls %5s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf gpurun_out/_tr
