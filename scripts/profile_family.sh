#!/bin/bash
# rocprofv3 kernel stats + two PMC passes for one parameter family through scripts/family_bench.py
# usage: scripts/profile_family.sh <param name> <count> <tag> [variant]; outputs under gpurun_out/fam_<tag>/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fam_$3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/scripts/family_bench.py $1 $2 1 ${4:-0}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $OUT/pmc_sq1 -o pmc -- $CMD > $OUT/pmc_sq1.log 2>&1 || { echo pmc1 failed; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1 || { echo pmc2 failed; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1 || { echo fetch failed; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1 || { echo write failed; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_l2 -o pmc -- $CMD > $OUT/pmc_l2.log 2>&1 || { echo l2 failed; }
find $OUT -type f \( -name "*.db" -o -name "*.json" -o -name "*.pftrace" \) -delete
find $OUT -type f -size +4M -delete
echo done $3
