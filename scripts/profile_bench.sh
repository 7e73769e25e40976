#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the bench command, then the two HBM-traffic PMC passes
# (separate runs, our kernels only, 2 denoising steps: traffic per launch does not depend on the step count).
# usage: scripts/profile_bench.sh <tag>      -> gpurun_out/<tag>_{stats,fetch,write}/
set -e -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:-prof}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- \
    python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_stats.json 2> $OUT/${TAG}_stats.err
echo "stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex pgn_ --output-format csv -d $OUT/${TAG}_fetch -- \
    python3 bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/${TAG}_fetch.err
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex pgn_ --output-format csv -d $OUT/${TAG}_write -- \
    python3 bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/${TAG}_write.err
echo "write done"
python3 scripts/profile_summary.py stats $OUT/${TAG}_stats $OUT/${TAG}_kernel_stats_top.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline" > /dev/null
python3 scripts/profile_summary.py pmc $OUT/${TAG}_pmc.json $OUT/${TAG}_fetch $OUT/${TAG}_write > /dev/null
# keep the merged-back payload small: the raw per-dispatch traces stay on the box
find $OUT/${TAG}_stats -name '*kernel_trace.csv' -delete
find $OUT/${TAG}_fetch $OUT/${TAG}_write -name '*counter_collection.csv' -delete
echo "summaries done"
