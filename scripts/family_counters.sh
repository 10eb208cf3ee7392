#!/bin/bash
# LDS / vector-ALU counters of one family_bench.py run: scripts/family_counters.sh <tag> <param name> <count> [variant]
# (set CONCRETE_AMD_LIB beforehand to compare builds); output under gpurun_out/famc_<tag>/
set -u
TAG=$1; NAME=$2; COUNT=$3; VAR=${4:-3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/famc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $OUT/pmc -o pmc -- python3 $ROOT/scripts/family_bench.py $NAME $COUNT 1 $VAR > $OUT/pmc.log 2>&1 || { echo pmc failed; tail -5 $OUT/pmc.log; exit 1; }
find $OUT -type f \( -name "*.db" -o -name "*.json" -o -name "*.pftrace" \) -delete
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "block_pbs" in k:
        print(k, {c: "%.4g" % x for c, x in sorted(v.items())})
PY
