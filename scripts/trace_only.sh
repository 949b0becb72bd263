#!/bin/bash
# kernel trace + stats of the default bench command for a few workloads (no counter passes): per-kernel average
# times of the REPLAYED sequence.  usage (GPU box): bash scripts/trace_only.sh tag "scircuit mac_econ"
set -u
TAG=${1:-t}
WORKLOADS=${2:-"scircuit mac_econ"}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
for w in $WORKLOADS; do
  rm -rf gpurun_out/_p_$w
  rocprofv3 --kernel-trace --stats -d gpurun_out/_p_$w/trace -o r -- python bench.py --workload $w --no-cpu-baseline --no-config5 --no-configs --no-verify ${BENCH_ARGS:-} \
      > $OUT/${w}_under_rocprof.log 2>&1
  python scripts/rocpd_summary.py $(find gpurun_out/_p_$w/trace -name "*.db" | head -1) $OUT/${w}_kernel_stats.csv > /dev/null
  echo "== $w"; head -14 $OUT/${w}_kernel_stats.csv | cut -c1-150
done
rm -rf gpurun_out/_p_*
