#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/profiles/ (copy the summaries into profiles/):
#   kernel-trace + stats of the default bench command, and separate FETCH_SIZE / WRITE_SIZE passes.
# usage (GPU box): bash scripts/collect_profiles.sh r01 "scircuit cant"
set -u
ROUND=${1:-r01}
WORKLOADS=${2:-"scircuit cant"}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/profiles
mkdir -p $OUT
for w in $WORKLOADS; do
  rm -rf gpurun_out/_p_$w
  rocprofv3 --kernel-trace --stats -d gpurun_out/_p_$w/trace -o r -- python bench.py --workload $w --no-cpu-baseline --no-config5 --no-configs --no-verify \
      > $OUT/${ROUND}_bench_${w}_under_rocprof.log 2>&1
  python scripts/rocpd_summary.py $(find gpurun_out/_p_$w/trace -name "*.db" | head -1) $OUT/${ROUND}_bench_${w}_kernel_stats.csv > /dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d gpurun_out/_p_$w/$c -o r -- python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-config5 --no-configs --no-verify \
        > gpurun_out/_p_$w/$c.log 2>&1
    python scripts/rocpd_pmc.py $(find gpurun_out/_p_$w/$c -name "*.db" | head -1) gpurun_out/_p_$w/$c.csv > /dev/null
  done
  python scripts/make_traffic.py gpurun_out/_p_$w/FETCH_SIZE.csv gpurun_out/_p_$w/WRITE_SIZE.csv \
      $OUT/${ROUND}_pmc_${w}_fetch_write.csv $OUT/traffic.json $w
  tail -1 $OUT/${ROUND}_bench_${w}_under_rocprof.log | cut -c1-400
done
ls -la $OUT
