#!/bin/bash
# scripts/abp.sh <rounds> <count> <params>... -- <lib>...: alternating throughput probes of several
# library builds on one box, for each named parameter set (e.g. PARAM_MESSAGE_4_CARRY_4)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; COUNT=$2; shift 2
PARAMS=()
while [ "$1" != "--" ]; do PARAMS+=("$1"); shift; done
shift
cd $ROOT
for p in "${PARAMS[@]}"; do
  for r in $(seq 1 $R); do
    for lib in "$@"; do
      echo -n "$p $lib: "
      QB_PARAMS=$p CONCRETE_AMD_LIB=$ROOT/$lib timeout -k 10 170 python scripts/quick_bench.py 0 $COUNT 3 | head -1 | sed 's/.*kernel pbs/pbs/' || exit 1
    done
  done
done
