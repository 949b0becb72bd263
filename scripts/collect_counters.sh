#!/bin/bash
# Regenerates the rocprofv3 evidence of a round under gpurun_out/profiles/ (copy the summaries into profiles/):
#   kernel-trace + stats of the default bench command per workload, and SEPARATE --pmc passes (kernel-trace only,
#   as gpurun requires): FETCH_SIZE | WRITE_SIZE | SQ / LDS | TCP.
# usage (GPU box): bash scripts/collect_counters.sh r05 "scircuit mac_econ cant webbase mac_econ_f32"
#   (a workload named <w>_f32 runs `bench.py --workload <w> --dtype f32`; its counters land under the key <w>_f32)
set -u
ROUND=${1:-r05}
WORKLOADS=${2:-"scircuit mac_econ cant webbase"}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
OUT=gpurun_out/profiles
mkdir -p $OUT
cp profiles/traffic.json $OUT/traffic.json 2>/dev/null
cp profiles/counters.json $OUT/counters.json 2>/dev/null
PASSES=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum"
 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
)
for w in $WORKLOADS; do
  base=${w%_f32}
  BENCH="python bench.py --no-cpu-baseline --no-lib-baseline --no-config5 --no-configs --no-verify --workload $base --detail gpurun_out/_p_$w/detail.json"
  if [ "$base" != "$w" ]; then BENCH="$BENCH --dtype f32"; fi
  rm -rf gpurun_out/_p_$w
  mkdir -p gpurun_out/_p_$w
  rocprofv3 --kernel-trace --stats -d gpurun_out/_p_$w/trace -o r -- $BENCH \
      > $OUT/${ROUND}_bench_${w}_under_rocprof.log 2>&1
  python scripts/rocpd_summary.py $(find gpurun_out/_p_$w/trace -name "*.db" | head -1) $OUT/${ROUND}_bench_${w}_kernel_stats.csv > /dev/null
  i=0
  CSVS=""
  for p in "${PASSES[@]}"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $p -d gpurun_out/_p_$w/pass$i -o r -- $BENCH --steps 5 --warmup 2 \
        > gpurun_out/_p_$w/pass$i.log 2>&1
    db=$(find gpurun_out/_p_$w/pass$i -name "*.db" | head -n 1)
    if [ -n "$db" ]; then python scripts/rocpd_pmc.py $db gpurun_out/_p_$w/pass$i.csv > /dev/null; CSVS="$CSVS gpurun_out/_p_$w/pass$i.csv"
    else echo "$w pass $i: no db"; tail -n 3 gpurun_out/_p_$w/pass$i.log; fi
  done
  python scripts/make_counters.py $w $OUT/${ROUND}_pmc_${w}_counters.csv $OUT/traffic.json $OUT/counters.json $CSVS
  tail -1 $OUT/${ROUND}_bench_${w}_under_rocprof.log | cut -c1-300
done
rm -rf gpurun_out/_p_*
ls -la $OUT
