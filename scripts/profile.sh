#!/bin/bash
# Collects the rocprofv3 kernel trace and PMC passes for bench.py on the GPU box.
# usage: scripts/profile.sh <tag> [bench.py arguments, e.g. --workload 4_4]   (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r1}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-buffer-pass $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $OUT/pmc_sq1 -o pmc -- $BENCH > $OUT/pmc_sq1.log 2>&1 || { echo pmc1 failed; tail -5 $OUT/pmc_sq1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1 || { echo pmc2 failed; tail -5 $OUT/pmc_sq2.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_I8 -d $OUT/pmc_f64 -o pmc -- $BENCH > $OUT/pmc_f64.log 2>&1 || { echo pmc_f64 failed; tail -5 $OUT/pmc_f64.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo write failed; tail -5 $OUT/pmc_write.log; exit 1; }
timeout -s KILL 200 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_l2 -o pmc -- $BENCH > $OUT/pmc_l2.log 2>&1 || { echo l2 failed; tail -5 $OUT/pmc_l2.log; }
# keep the summaries only: databases and per-call traces stay on the box
find $OUT -type f \( -name "*.db" -o -name "*.json" -o -name "*.pftrace" \) -delete
find $OUT -type f -size +4M -delete
find $OUT -type f | xargs ls -la | head -40
