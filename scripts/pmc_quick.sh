#!/bin/bash
# scripts/pmc_quick.sh <tag> <count> <counters...>: one rocprofv3 PMC pass over scripts/quick_bench.py
# (2_2, device-resident random inputs, one step); environment (CAMD_WAVE_SCHEDULE ...) is inherited.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; CNT=$2; shift 2
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --output-format csv --pmc "$@" -d $OUT/a -o pmc -- python3 $ROOT/scripts/quick_bench.py ${CAMD_VARIANT:-0} $CNT 1 > $OUT/a.log 2>&1 || { echo pass failed; tail -5 $OUT/a.log; exit 1; }
find $OUT -type f \( -name "*.db" -o -name "*.json" \) -delete
python3 - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1]+'/*/pmc_counter_collection.csv'):
    for row in csv.DictReader(open(p)):
        k=row['Kernel_Name']
        if 'pbs_kernel' in k or 'keyswitch' in k:
            acc[k[:70]][row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()): print(f"   {c:28s} {sum(vals)/len(vals):.5g}")
PY
