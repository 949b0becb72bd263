cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
mkdir -p gpurun_out/s5
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/s5/t_full.log 2>&1; echo "tests rc=$?" >> gpurun_out/s5/t_full.log
tail -4 gpurun_out/s5/t_full.log
run() { tag=$1; shift; timeout 900 python tests/tools/stress_gpu.py "$@" > gpurun_out/s5/stress_$tag.log 2>&1; echo "$tag rc=$? $(tail -n 1 gpurun_out/s5/stress_$tag.log | cut -c1-200)"; }
run v2_single 300 5001 num_verify=2
run v2_turns 1000 5101 interleave=4 num_verify=2
run v1_turns 600 5102 interleave=3
