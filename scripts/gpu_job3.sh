#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
mkdir -p gpurun_out/j3
( time timeout 900 python bench.py ) > gpurun_out/j3/bench_default.log 2>&1
SPECK_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --config5-scale 0.01 > gpurun_out/j3/bench_shared2.log 2>&1
SPECK_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 5 --warmup 2 --workload nlpkkt --scale 0.01 --scaling strong > gpurun_out/j3/bench_shared2_strong.log 2>&1
tail -n 5 gpurun_out/j3/*.log
