#!/bin/bash
# scripts/abn.sh <rounds> <lib>...: alternating throughput probes of several library builds on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; shift
cd $ROOT
for r in $(seq 1 $R); do
  for lib in "$@"; do
    echo -n "$lib: "
    CONCRETE_AMD_LIB=$ROOT/$lib timeout -k 10 120 python scripts/quick_bench.py 0 16384 3 | head -1 | sed 's/.*kernel pbs/pbs/' || exit 1
  done
done
