#!/bin/bash
# scripts/ab.sh <libA> <libB> [rounds]: alternating throughput probes of two library builds on one box
# (paths relative to the repository root; environment such as CAMD_WAVE_SCHEDULE is inherited).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; R=${3:-2}
cd $ROOT
for r in $(seq 1 $R); do
  for lib in $A $B; do
    echo -n "$lib: "
    CONCRETE_AMD_LIB=$ROOT/$lib timeout -k 10 120 python scripts/quick_bench.py 0 16384 3 | head -1 || exit 1
  done
done
