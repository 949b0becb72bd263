#!/bin/bash
# Regenerates every measured artefact of a round on the GPU box (results under gpurun_out/profiles/, copy them
# into profiles/):  bash scripts/refresh_round.sh r05
ROUND=${1:-r06}
WORKLOADS=${2:-"scircuit mac_econ cant webbase nlpkkt mac_econ_f32 cant_f32"}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
OUT=gpurun_out/profiles
mkdir -p $OUT
bash scripts/collect_counters.sh $ROUND "$WORKLOADS" > $OUT/collect.log 2>&1
cp $OUT/counters.json $OUT/traffic.json profiles/   # the bench lines read the ceilings of THIS round's passes
for w in scircuit mac_econ cant webbase uniform nlpkkt; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-lib-baseline --no-config5 --no-configs --detail $OUT/${ROUND}_bench_${w}_detail.json 2> /dev/null | tail -n 1 > $OUT/${ROUND}_bench_$w.json
done
# the launch durations of the plain lines against the traces of the same commands under rocprofv3 (<= 5 %)
for w in scircuit mac_econ cant webbase nlpkkt; do
  python scripts/check_launch_ms.py $OUT/${ROUND}_bench_${w}_detail.json $OUT/${ROUND}_bench_${w}_kernel_stats.csv > $OUT/${ROUND}_launch_ms_check_$w.txt 2>&1 \
      || echo "$w: launch ms differ from the trace by more than 5 %" >> $OUT/${ROUND}_launch_ms_check_$w.txt
done
timeout 900 python bench.py --detail $OUT/${ROUND}_bench_default_detail.json 2> /dev/null | tail -n 1 > $OUT/${ROUND}_bench_default.json
wc -c $OUT/${ROUND}_bench_default.json
# the ONE-WALK complete call (option one_walk, off by default: measured and lost): line + kernel trace per input
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in scircuit mac_econ cant; do
  rm -rf gpurun_out/_ow
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/_ow -o r -- python bench.py --workload $w --opt one_walk=2 --no-cpu-baseline --no-lib-baseline --no-config5 --no-configs --no-reuse --detail gpurun_out/_ow/detail.json 2> /dev/null | tail -n 1 > $OUT/${ROUND}_onewalk_${w}.json
  python scripts/rocpd_summary.py $(find gpurun_out/_ow -name "*.db" | head -1) $OUT/${ROUND}_onewalk_${w}_kernel_stats.csv > /dev/null
done
rm -rf gpurun_out/_ow
bash scripts/timeline.sh ${ROUND}_onewalk_scircuit scircuit 12 --no-reuse --no-lib-baseline --opt one_walk=2 > /dev/null 2>&1; cp gpurun_out/timeline/${ROUND}_onewalk_scircuit.txt $OUT/
bash scripts/timeline.sh ${ROUND}_complete_scircuit scircuit 10 --no-reuse --no-lib-baseline > /dev/null 2>&1; cp gpurun_out/timeline/${ROUND}_complete_scircuit.txt $OUT/
# ... and with the through call off (option eager_through=0: the read-back between scan and numeric launches)
bash scripts/timeline.sh ${ROUND}_complete_nothrough_scircuit scircuit 10 --no-reuse --no-lib-baseline --opt eager_through=0 > /dev/null 2>&1; cp gpurun_out/timeline/${ROUND}_complete_nothrough_scircuit.txt $OUT/
timeout 300 python scripts/multiwindow_time.py 2>&1 | grep windows > $OUT/${ROUND}_multiwindow_now.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/profiles/*_bench_*.json")):
    if f.endswith("_detail.json"):
        continue
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d.get("roofline") or {}
    print(f.split("/")[-1], "ms", d["ms_per_step"], "reuse", d.get("ms_reuse"), "GFLOP/s", d["value"], d["phases_ms"], r.get("kernel"),
          "frac", r.get("frac"), "bound", r.get("bound"), "phase", r.get("numeric_phase_frac"), "traffic", r.get("traffic"),
          [(c["name"], c["dtype"], c["ms_per_step"], c["ms_reuse"], c["roofline_frac"], c["bound"]) for c in d.get("configs", [])],
          d.get("config5"), d.get("cpu_baseline"))
PY
