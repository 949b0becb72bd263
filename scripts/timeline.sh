#!/bin/bash
# start / duration / end of the last kernels of a bench run (one replayed multiply = the tail): where the time between
# the launches of a replayed sequence goes.  usage (GPU box): bash scripts/timeline.sh tag workload [n] [bench opts...]
TAG=$1; W=$2; N=${3:-24}; shift 3
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
mkdir -p gpurun_out/timeline
rm -rf gpurun_out/_tl_tmp
rocprofv3 --kernel-trace -d gpurun_out/_tl_tmp -o r -- python bench.py --workload $W --steps 5 --warmup 3 --no-cpu-baseline --no-config5 --no-configs --no-verify "$@" > gpurun_out/timeline/${TAG}.log 2>&1
db=$(find gpurun_out/_tl_tmp -name "*.db" | head -n 1)
python scripts/rocpd_timeline.py $db $N > gpurun_out/timeline/${TAG}.txt
rm -rf gpurun_out/_tl_tmp
cat gpurun_out/timeline/${TAG}.txt
