#!/bin/bash
# usage (GPU box): bash scripts/pmc_passes.sh <tag> <workload> [bench opts...]   -> gpurun_out/pmc/<tag>_passN.csv
# separate rocprofv3 --pmc passes (kernel-trace only, as gpurun requires) of a short bench run
TAG=$1; W=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
mkdir -p gpurun_out/pmc
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
 "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum"
)
i=0
for p in "${PASSES[@]}"; do
  i=$((i+1))
  rm -rf gpurun_out/_pmc_tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $p -d gpurun_out/_pmc_tmp -o r -- python bench.py --workload $W --steps 3 --warmup 2 --no-cpu-baseline --no-config5 --no-configs --no-verify "$@" > gpurun_out/pmc/${TAG}_pass$i.log 2>&1
  db=$(find gpurun_out/_pmc_tmp -name "*.db" | head -n 1)
  if [ -n "$db" ]; then python scripts/rocpd_pmc.py $db gpurun_out/pmc/${TAG}_pass$i.csv > /dev/null; else echo "pass $i: no db"; tail -n 3 gpurun_out/pmc/${TAG}_pass$i.log; fi
done
rm -rf gpurun_out/_pmc_tmp
