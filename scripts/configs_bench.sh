#!/bin/bash
# One-GPU shares of the other BASELINE configs and the small-batch points (run through gpurun); one JSON line each.
set -o pipefail
cd "$(dirname "$0")/.."
run() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 420 python bench.py --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/cfg_$tag.json 2> gpurun_out/cfg_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/cfg_$tag.err; return 1; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/cfg_$tag.json").read().strip().splitlines()[-1])
print("$tag", "images/s", round(d["value"],4), "ms/step", round(d["ms_per_step"],1), "loop-only images/s", round(d["loop_only"]["images_per_sec_per_gpu"],4))
PY
}
run b1 --batch 1 --steps 3 &&
run b4 --batch 4 --steps 2 &&
run cfg3_share --batch 8 --height 1216 --width 832 --denoise-steps 50 --steps 1 &&
run cfg4_share --batch 16 --num-return-images 8 --steps 1 &&
run b16 --steps 2
