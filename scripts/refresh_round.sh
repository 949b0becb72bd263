#!/bin/bash
# Regenerates every measured artefact of a round on the GPU box (results under gpurun_out/profiles/, copy them
# into profiles/):  bash scripts/refresh_round.sh r02
ROUND=${1:-r02}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
OUT=gpurun_out/profiles
mkdir -p $OUT
bash scripts/collect_counters.sh $ROUND "scircuit mac_econ cant webbase nlpkkt" > $OUT/collect.log 2>&1
cp $OUT/counters.json $OUT/traffic.json profiles/   # the bench lines read the ceilings of THIS round's passes
for w in scircuit mac_econ cant webbase uniform nlpkkt; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-config5 --no-configs 2> /dev/null | tail -n 1 > $OUT/${ROUND}_bench_$w.json
done
# the launch durations of the plain lines against the traces of the same commands under rocprofv3 (<= 5 %)
for w in scircuit mac_econ cant webbase nlpkkt; do
  python scripts/check_launch_ms.py $OUT/${ROUND}_bench_$w.json $OUT/${ROUND}_bench_${w}_kernel_stats.csv > $OUT/${ROUND}_launch_ms_check_$w.txt 2>&1 \
      || echo "$w: launch ms differ from the trace by more than 5 %" >> $OUT/${ROUND}_launch_ms_check_$w.txt
done
timeout 900 python bench.py 2> /dev/null | tail -n 1 > $OUT/${ROUND}_bench_default.json
timeout 300 python scripts/multiwindow_time.py 2>&1 | grep windows > $OUT/${ROUND}_multiwindow_now.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/profiles/*_bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d.get("roofline") or {}
    print(f.split("/")[-1], d["ms_per_step"], d["value"], d["phases_ms"]["symbolic"], d["phases_ms"]["numeric"], r.get("kernel"), r.get("frac"),
          "phase", r.get("numeric_phase_frac"), "traffic", r.get("traffic"), json.dumps(r.get("launches")), d.get("config5", {}).get("value"), d.get("cpu_baseline"))
PY
