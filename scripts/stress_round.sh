#!/bin/bash
# the randomised stress of a round (tests/tools/stress_gpu.py): single problems and problems taking turns on one config,
# default options and a few option sets; since round 6 also under CANARY ZONES (SPECK_GUARD_BYTES: every device buffer of
# the library carries zones that are compared after every call, speck_amd/csrc/guards.hpp), with the one-walk call on,
# and the host side under AddressSanitizer / UBSan (scripts/asan_suite.sh).  usage (GPU box): bash scripts/stress_round.sh r06
R=${1:-r06}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
mkdir -p gpurun_out/stress
run() { tag=$1; shift; timeout 1500 python tests/tools/stress_gpu.py "$@" > gpurun_out/stress/${R}_$tag.log 2>&1; echo "$tag rc=$? $(tail -n 1 gpurun_out/stress/${R}_$tag.log | cut -c1-200)"; }
run single_a 400 4001
run single_b 300 4002 esc32=0
run single_c 300 4003 overlap_analysis=0 eager_speculate=0
run single_d 300 4004 reuse=0
run single_w 300 4005 reuse=0 one_walk=2
run single_s 300 4006 slice_rows=1
run turns_a 1500 4101 interleave=4
run turns_b 1500 4102 interleave=3 esc64=0
run turns_c 1000 4103 interleave=5 nf_min_ops=1
run turns_w 1000 4104 interleave=4 reuse=0 one_walk=2
STRESS_REPEAT=4 run repeat_a 500 7101 interleave=3 num_verify=2
STRESS_REPEAT=4 run repeat_b 400 7103 interleave=3
STRESS_HOSTILE=1 STRESS_REPEAT=3 run hostile_a 600 8101 interleave=3
STRESS_HOSTILE=1 run hostile_b 1000 8102 interleave=4 num_verify=2
STRESS_HOSTILE=1 STRESS_REPEAT=2 run hostile_c 500 8103 interleave=3 esc_fused=0
STRESS_HOSTILE=1 STRESS_REPEAT=4 run hostile_d 800 8104 interleave=5
STRESS_HOSTILE=1 STRESS_REPEAT=2 run hostile_w 600 8105 interleave=3 reuse=0 one_walk=2
STRESS_HOSTILE=1 STRESS_REPEAT=2 run hostile_s 600 8106 interleave=3 slice_rows=1
# ... and under canary zones: a touched zone turns a call into SPECK_ERR_HIP, which the tool counts as a failure
export SPECK_GUARD_BYTES=4096
run guard_single 300 9001
run guard_turns 800 9002 interleave=4
STRESS_HOSTILE=1 STRESS_REPEAT=3 run guard_hostile_a 600 9003 interleave=3
STRESS_HOSTILE=1 run guard_hostile_b 600 9004 interleave=4 num_verify=2
STRESS_HOSTILE=1 STRESS_REPEAT=2 run guard_hostile_w 500 9005 interleave=3 reuse=0 one_walk=2
STRESS_HOSTILE=1 STRESS_REPEAT=2 run guard_hostile_s 500 9006 interleave=3 slice_rows=1
(time python -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_guards.py) > gpurun_out/stress/${R}_guard_full_suite.log 2>&1
echo "guard_full_suite rc=$? $(tail -n 4 gpurun_out/stress/${R}_guard_full_suite.log | head -n 1)"
unset SPECK_GUARD_BYTES
bash scripts/asan_suite.sh gpu > gpurun_out/stress/${R}_asan_gpu.log 2>&1; echo "asan_gpu rc=$? $(tail -n 1 gpurun_out/stress/${R}_asan_gpu.log)"
bash scripts/asan_suite.sh cpu > gpurun_out/stress/${R}_asan_cpu.log 2>&1; echo "asan_cpu rc=$? $(tail -n 1 gpurun_out/stress/${R}_asan_cpu.log)"
