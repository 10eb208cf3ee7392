#!/bin/bash
# scripts/ab_tree.sh <rounds> <count>: the 2_2 throughput probe of this tree against that of another
# tree built under build/oldtree (its own package and library), alternating on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq 1 $1); do
  for tree in $ROOT/build/oldtree $ROOT; do
    echo -n "$(basename $tree): "
    (cd $tree && timeout -k 10 170 python scripts/quick_bench.py 0 $2 3 | head -1 | sed 's/.*kernel pbs/pbs/') || exit 1
  done
done
