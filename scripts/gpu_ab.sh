#!/bin/bash
# A/B of library options on the four single-GPU workloads: bash scripts/gpu_ab.sh "optA=1" "optB=0 optC=1" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
mkdir -p gpurun_out/ab
WL=${WL:-"scircuit mac_econ cant webbase"}
for w in $WL; do
  for variant in "$@"; do
    opts=""
    for o in $variant; do [ "$o" != "base" ] && opts="$opts --opt $o"; done
    timeout 600 python bench.py --workload $w --no-cpu-baseline --no-config5 --no-configs --no-verify $BENCH_ARGS $opts 2>&1 | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-9s %-34s %.4f ms  eager %.4f  %7.1f GF  sym %.3f num %.3f  phase %.3f  %s' % ('$w', '$variant', d['ms_per_step'], d['eager_ms_per_step'] or 0, d['value'], d['phases_ms']['symbolic'], d['phases_ms']['numeric'], d['roofline']['numeric_phase_frac'], d['kernels_ms']))
"
  done
done
