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
print("$tag", "images/s", round(d["value"],4), "ms/step", round(d["ms_per_step"],1), "loop-only images/s", (round(d["loop_only"]["images_per_sec_per_gpu"],4) if (d.get("loop_only") or {}).get("images_per_sec_per_gpu") else None))
PY
}
run b1 --batch 1 --steps 3 &&
run b2 --batch 2 --steps 3 &&
run b4 --batch 4 --steps 2 &&
run b1_graph --batch 1 --steps 3 --hip-graph &&
run config2_share --config 2 --steps 1 &&
run config3_share --config 3 --steps 1 &&
run config4 --config 4 --steps 2 &&
run config1 --steps 2
