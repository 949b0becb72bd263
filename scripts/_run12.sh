cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; mkdir -p gpurun_out/profiles
timeout 900 python bench.py --workload nlpkkt --no-cpu-baseline --no-config5 --no-configs 2>/dev/null | tail -n 1 > gpurun_out/profiles/r04_bench_nlpkkt.json
python scripts/check_launch_ms.py gpurun_out/profiles/r04_bench_nlpkkt.json profiles/r04_bench_nlpkkt_kernel_stats.csv > gpurun_out/profiles/r04_launch_ms_check_nlpkkt.txt 2>&1
cat gpurun_out/profiles/r04_launch_ms_check_nlpkkt.txt
python -c "
import json; d=json.load(open('gpurun_out/profiles/r04_bench_nlpkkt.json')); print(d['ms_per_step'], d['value'], d['phases_ms']['symbolic'], d['phases_ms']['numeric'], d['roofline']['frac'], d['roofline']['numeric_phase_frac'], d['verified'])"
bash scripts/stress_round.sh r04
