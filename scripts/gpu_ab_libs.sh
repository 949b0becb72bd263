#!/bin/bash
# A/B of whole library builds: build each variant with `make` and copy speck_amd/libspeck_amd.so to
# speck_amd/variants/<name>.so, then on the GPU box:  bash scripts/gpu_ab_libs.sh name1 name2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
WL=${WL:-"scircuit mac_econ cant webbase"}
cp speck_amd/libspeck_amd.so /tmp/lib_keep.so
for w in $WL; do
  for v in "$@"; do
    cp speck_amd/variants/$v.so speck_amd/libspeck_amd.so
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-config5 --no-configs --no-verify $BENCH_ARGS 2>&1 | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-9s %-20s %.4f ms  %7.1f GF  sym %.3f num %.3f  %s' % ('$w', '$v', d['ms_per_step'], d['value'], d['phases_ms']['symbolic'], d['phases_ms']['numeric'], d['kernels_ms']))
"
  done
done
cp /tmp/lib_keep.so speck_amd/libspeck_amd.so
