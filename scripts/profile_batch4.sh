#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_batch4_stats -- python3 bench.py --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/r03_batch4_stats.json 2> $OUT/r03_batch4_stats.err
python3 scripts/profile_summary.py stats $OUT/r03_batch4_stats $OUT/r03_batch4_kernel_stats_top.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing" > /dev/null
find $OUT/r03_batch4_stats -name '*kernel_trace.csv' -delete
head -45 $OUT/r03_batch4_kernel_stats_top.txt | cut -c1-70,110-190
