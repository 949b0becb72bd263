cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
mkdir -p gpurun_out/s5
run() { tag=$1; shift; timeout 1200 python tests/tools/stress_gpu.py "$@" > gpurun_out/s5/stress_$tag.log 2>&1; echo "$tag rc=$? $(tail -n 1 gpurun_out/s5/stress_$tag.log | cut -c1-200)"; }
run w_single 300 6001 num_verify=2
run w_turns 1200 6101 interleave=4 num_verify=2
run w_turns_def 800 6102 interleave=3
STRESS_SCALE=8 run w_big 60 6201 num_verify=2
STRESS_SCALE=8 run w_big_turns 150 6202 interleave=3 num_verify=2
grep -c "pred=31" gpurun_out/s5/stress_w_*.log
