#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
rm -f profiles/traffic.json
bash scripts/collect_profiles.sh r02 "scircuit mac_econ cant webbase" > gpurun_out/collect.log 2>&1
cp gpurun_out/profiles/traffic.json profiles/traffic.json
sed -i 's#"_source": "gpurun_out/profiles/#"_source": "profiles/#' profiles/traffic.json
cp profiles/traffic.json gpurun_out/profiles/traffic.json
python bench.py > gpurun_out/profiles/r02_bench_default.json 2> gpurun_out/bench_default.err
tail -c 300 gpurun_out/profiles/r02_bench_default.json
for w in scircuit mac_econ cant webbase; do python bench.py --workload $w --no-config5 --no-cpu-baseline 2>/dev/null | tail -n 1 > gpurun_out/profiles/r02_bench_${w}.json; done
