#!/bin/bash
# The whole GPU suite on a library whose kernels poison their LDS first (make POISON=1; device_common.hpp, poison_lds): a body
# that reads an LDS word it never wrote gathers at a wild address or fails its parity test.  usage (GPU box):
#   bash scripts/poison_suite.sh          -> gpurun_out/poison/suite.log
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONPATH=$PWD
mkdir -p gpurun_out/poison
make POISON=1 -j8 speck_amd/libspeck_amd_poison.so > gpurun_out/poison/build.log 2>&1 || { echo "POISON build failed"; tail gpurun_out/poison/build.log; exit 2; }
export SPECK_LIB=$PWD/speck_amd/libspeck_amd_poison.so
(time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider) > gpurun_out/poison/suite.log 2>&1
rc=$?
tail -n 6 gpurun_out/poison/suite.log
echo "poison suite rc=$rc"
exit $rc
