#!/bin/bash
# the randomised stress of a round (tests/tools/stress_gpu.py): single problems and problems taking turns on one config,
# default options and a few option sets.  usage (GPU box): bash scripts/stress_round.sh r04
R=${1:-r04}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
mkdir -p gpurun_out/stress
run() { tag=$1; shift; timeout 1500 python tests/tools/stress_gpu.py "$@" > gpurun_out/stress/${R}_$tag.log 2>&1; echo "$tag rc=$? $(tail -n 1 gpurun_out/stress/${R}_$tag.log | cut -c1-200)"; }
run single_a 400 4001
run single_b 300 4002 esc32=0
run single_c 300 4003 overlap_analysis=0 eager_speculate=0
run single_d 300 4004 reuse=0
run turns_a 1500 4101 interleave=4
run turns_b 1500 4102 interleave=3 esc64=0
run turns_c 1000 4103 interleave=5 nf_min_ops=1
STRESS_REPEAT=4 run repeat_a 500 7101 interleave=3 num_verify=2
STRESS_REPEAT=4 run repeat_b 400 7103 interleave=3
STRESS_HOSTILE=1 STRESS_REPEAT=3 run hostile_a 600 8101 interleave=3
STRESS_HOSTILE=1 run hostile_b 1000 8102 interleave=4 num_verify=2
STRESS_HOSTILE=1 STRESS_REPEAT=2 run hostile_c 500 8103 interleave=3 esc_fused=0
STRESS_HOSTILE=1 STRESS_REPEAT=4 run hostile_d 800 8104 interleave=5
