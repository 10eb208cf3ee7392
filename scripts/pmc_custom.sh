#!/bin/bash
# scripts/pmc_custom.sh <param> <count> <tag> <counters...>: one PMC pass with the given counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
P=$1; CNT=$2; TAG=$3; shift 3
OUT=$ROOT/gpurun_out/pmcc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --output-format csv --pmc "$@" -d $OUT/a -o pmc -- python3 $ROOT/scripts/family_bench.py $P $CNT 1 ${CAMD_VARIANT:-0} > $OUT/a.log 2>&1 || { echo pass failed; tail -5 $OUT/a.log; exit 1; }
find $OUT -type f \( -name "*.db" -o -name "*.json" \) -delete
python3 - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1]+'/*/pmc_counter_collection.csv'):
    for row in csv.DictReader(open(p)):
        k=row['Kernel_Name']
        if 'pbs_kernel' in k:
            acc[k[:60]][row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()): print(f"   {c:24s} {sum(vals)/len(vals):.4g}")
PY
