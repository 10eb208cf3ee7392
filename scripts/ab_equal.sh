#!/bin/bash
# scripts/ab_equal.sh <libA> <libB> <param> <count> [<param> <count> ...]: output digests of two library builds
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; shift 2
cd $ROOT
while [ $# -ge 2 ]; do
  for lib in $A $B; do
    echo -n "$lib: "
    CONCRETE_AMD_LIB=$ROOT/$lib timeout -k 10 170 python scripts/output_hash.py $1 $2 || exit 1
  done
  shift 2
done
