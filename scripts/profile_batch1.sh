#!/bin/bash
# Batch-1 operating point: is a step launch-bound or kernel-bound?  Kernel-trace stats of 3 steps (1 warm-up + 2 timed)
# against the wall time of the bench line.   usage: scripts/profile_batch1.sh <tag> [batch]
set -e -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:-b1}
B=${2:-1}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- \
    python3 bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/${TAG}_stats.json 2> $OUT/${TAG}_stats.err
python3 scripts/profile_summary.py stats $OUT/${TAG}_stats $OUT/${TAG}_kernel_stats_top.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing" > /dev/null
find $OUT/${TAG}_stats -name '*kernel_trace.csv' -delete
timeout -s KILL 300 python3 bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/${TAG}_plain.json
cat $OUT/${TAG}_stats.json $OUT/${TAG}_plain.json
tail -n 1 $OUT/${TAG}_kernel_stats_top.txt
