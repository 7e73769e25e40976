#!/bin/bash
# End-to-end A/B of the one-pass GroupNorm rule at small batches: bench.py with the default rule against
# PGN_GN_NHWC_ONEPASS=0, alternating, two rounds, one box (run through gpurun).
set -o pipefail
cd "$(dirname "$0")/.."
one() {  # tag, env value or "", bench args...
  tag=$1; sw=$2; shift; shift
  if [ -n "$sw" ]; then export PGN_GN_NHWC_ONEPASS=$sw; else unset PGN_GN_NHWC_ONEPASS; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/gn_e2e_$tag.json 2> gpurun_out/gn_e2e_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/gn_e2e_$tag.err; return 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/gn_e2e_$tag.json').read().strip().splitlines()[-1])
print('$tag', 'images/s', round(d['value'],4), 'ms/step', round(d['ms_per_step'],1))"
}
for r in 1 2; do
  one b1_auto_$r "" --batch 1 --steps 4 &&
  one b1_three_$r 0 --batch 1 --steps 4 &&
  one b2_auto_$r "" --batch 2 --steps 3 &&
  one b2_three_$r 0 --batch 2 --steps 3 &&
  one b4_auto_$r "" --batch 4 --steps 2 &&
  one b4_three_$r 0 --batch 4 --steps 2 || exit 1
done
