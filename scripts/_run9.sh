cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
mkdir -p gpurun_out/s5
run() { tag=$1; shift; timeout 1500 python tests/tools/stress_gpu.py "$@" > gpurun_out/s5/stress_$tag.log 2>&1; echo "$tag rc=$? $(tail -n 1 gpurun_out/s5/stress_$tag.log | cut -c1-200) pred31=$(grep -c pred=31 gpurun_out/s5/stress_$tag.log)"; }
STRESS_REPEAT=4 run r_turns 500 7101 interleave=3 num_verify=2
STRESS_REPEAT=5 run r_turns_b 400 7102 interleave=2 num_verify=2
STRESS_REPEAT=4 run r_turns_def 400 7103 interleave=3
STRESS_REPEAT=4 STRESS_SCALE=8 run r_big 80 7201 interleave=2 num_verify=2
