#!/bin/bash
# one PMC pass over scripts/family_bench.py: scripts/pmc_family.sh <param> <count> <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcfam_$3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/a -o pmc -- python3 $ROOT/scripts/family_bench.py $1 $2 1 ${CAMD_VARIANT:-0} > $OUT/a.log 2>&1 || { echo pass a failed; tail -5 $OUT/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT -d $OUT/b -o pmc -- python3 $ROOT/scripts/family_bench.py $1 $2 1 ${CAMD_VARIANT:-0} > $OUT/b.log 2>&1 || { echo pass b failed; tail -5 $OUT/b.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/c -o pmc -- python3 $ROOT/scripts/family_bench.py $1 $2 1 ${CAMD_VARIANT:-0} > $OUT/c.log 2>&1 || { echo pass c failed; tail -5 $OUT/c.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/d -o pmc -- python3 $ROOT/scripts/family_bench.py $1 $2 1 ${CAMD_VARIANT:-0} > $OUT/d.log 2>&1 || { echo pass d failed; tail -5 $OUT/d.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $OUT/e -o pmc -- python3 $ROOT/scripts/family_bench.py $1 $2 1 ${CAMD_VARIANT:-0} > $OUT/e.log 2>&1 || { echo pass e failed; tail -5 $OUT/e.log; }
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
