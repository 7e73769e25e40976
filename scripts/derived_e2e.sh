#!/bin/bash
# End-to-end A/B of PGN_KEEP_DERIVED at small batches (bench.py, alternating, two rounds, one box; run through gpurun).
set -o pipefail
cd "$(dirname "$0")/.."
one() {  # tag, switch value, bench args...
  tag=$1; sw=$2; shift; shift
  PGN_KEEP_DERIVED=$sw timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/drv_$tag.json 2> gpurun_out/drv_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/drv_$tag.err; return 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/drv_$tag.json').read().strip().splitlines()[-1])
print('$tag', 'images/s', round(d['value'],4), 'ms/step', round(d['ms_per_step'],1))"
}
for r in 1 2; do
  one b1_keep_$r 1 --batch 1 --steps 4 &&
  one b1_recompute_$r 0 --batch 1 --steps 4 &&
  one b2_keep_$r 1 --batch 2 --steps 3 &&
  one b2_recompute_$r 0 --batch 2 --steps 3 &&
  one b4_keep_$r 1 --batch 4 --steps 2 &&
  one b4_recompute_$r 0 --batch 4 --steps 2 || exit 1
done
