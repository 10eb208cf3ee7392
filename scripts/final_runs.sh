#!/bin/bash
# The bench lines and kernel traces that profiles/<round>/ keeps, taken on one box:
# scripts/final_runs.sh   (outputs under gpurun_out/final/)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err || { echo default bench failed; tail -5 $OUT/bench_final.err; exit 1; }
tail -c 600 $OUT/bench_final.json; echo
timeout -k 10 600 python bench.py --workload 4_4 > $OUT/bench_4_4_final.json 2> $OUT/bench_4_4.err || { echo 4_4 bench failed; tail -5 $OUT/bench_4_4.err; exit 1; }
tail -c 600 $OUT/bench_4_4_final.json; echo
timeout -k 10 300 python bench.py --workload boolean > $OUT/bench_boolean.json 2> $OUT/bench_boolean.err || { echo boolean bench failed; exit 1; }
timeout -k 10 300 python bench.py --workload 3_3 > $OUT/bench_3_3.json 2> $OUT/bench_3_3.err || { echo 3_3 bench failed; exit 1; }
timeout -k 10 400 python bench.py --workload fheuint8 > $OUT/bench_fheuint8.json 2> $OUT/bench_fheuint8.err || { echo fheuint8 bench failed; tail -5 $OUT/bench_fheuint8.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-buffer-pass > $OUT/trace.log 2>&1 || { echo trace failed; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace44 -o trace -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-buffer-pass --workload 4_4 > $OUT/trace44.log 2>&1 || { echo trace 4_4 failed; exit 1; }
find $OUT -type f \( -name "*.db" -o -name "*.pftrace" \) -delete
find $OUT -type f -size +4M -delete
grep -h pbs_kernel $OUT/trace/trace_kernel_stats.csv $OUT/trace44/trace_kernel_stats.csv | cut -c1-200
