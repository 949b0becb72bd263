cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
mkdir -p gpurun_out/s5
WL="webbase" timeout 900 bash scripts/gpu_ab.sh base "num_verify=0" "num_verify=2" > gpurun_out/s5/ab7.log 2>&1
cat gpurun_out/s5/ab7.log
timeout 600 python bench.py --workload webbase --no-cpu-baseline --no-config5 --no-configs --no-f32 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('webbase verified', d['ms_per_step'], d['verified'], d['verify'].get('pred_stages'), d['phases_ms']['symbolic'], d['phases_ms']['numeric'])"
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/s5/t_full.log 2>&1; echo "tests rc=$?" >> gpurun_out/s5/t_full.log
tail -4 gpurun_out/s5/t_full.log
