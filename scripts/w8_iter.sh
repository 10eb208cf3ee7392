#!/bin/bash
# One GPU iteration on the two-waves-per-SIMD schedule: parity (persistent rounds + noise variance at
# 2_2), then a throughput probe. usage: scripts/w8_iter.sh [schedule]
set -u
export CAMD_WAVE_SCHEDULE=${1:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 300 python -u -m pytest -x -q --timeout 200 --timeout-method thread tests/test_gpu_parity.py -m gpu \
  -k "persistent_rounds or noise_variance[PARAM_MESSAGE_2_CARRY_2]" -s > gpurun_out/w8_iter.log 2>&1
rc=$?
grep -E "variance|passed|failed|Error|assert" gpurun_out/w8_iter.log | head -8
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python scripts/quick_bench.py 0 16384 3
