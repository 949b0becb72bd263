cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; mkdir -p gpurun_out/s5
WL="scircuit mac_econ webbase" bash scripts/gpu_ab.sh base "keep_cols=0" base "keep_cols=0"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
