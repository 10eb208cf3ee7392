#!/bin/bash
# Throughput of the large-polynomial sets quoted in the kernel table (scripts/family_bench.py, random keys)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "PARAM_MESSAGE_3_CARRY_4 1024" "PARAM_MESSAGE_7_CARRY_0 1024" "PARAM_MESSAGE_3_CARRY_3 4096" "PARAM_MESSAGE_2_CARRY_4 4096" "PARAM_MESSAGE_2_CARRY_3 4096" "PARAM_MESSAGE_4_CARRY_4 512" "PARAM_MESSAGE_4_CARRY_4 1"; do
  timeout -k 10 170 python scripts/family_bench.py $spec 2 || exit 1
done
timeout -k 10 170 python scripts/family_bench.py PARAM_MESSAGE_1_CARRY_7 512 2 2 || exit 1
