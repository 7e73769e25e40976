#!/bin/bash
# Same-box A/B of the producer-side GroupNorm statistics (layers.PRODUCER_STATS): bench.py end to end, off / on / off / on.
# usage: scripts/producer_stats_ab.sh <tag>    -> gpurun_out/<tag>.txt
set -u
tag=${1:-r03_producer_stats_ab}
out=gpurun_out/$tag.txt
mkdir -p gpurun_out
: > "$out"
for sw in 0 1 0 1; do
    PGN_PRODUCER_STATS=$sw timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag.$sw.json 2> gpurun_out/$tag.$sw.err || exit 1
    python3 - "$sw" gpurun_out/$tag.$sw.json >> "$out" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["kernels"]
gn = {n: k[n] for n in k if n.startswith("groupnorm") or n == "bias_residual"}
print(f"PGN_PRODUCER_STATS={sys.argv[1]}: {d['value']:.4f} images/s, {d['ms_per_step']:.1f} ms/step, loop-only "
      f"{d['loop_only']['images_per_sec_per_gpu']:.4f}; " +
      "; ".join(f"{n}: {v['launches']} x {v['avg_us']:.1f} us = {v['share_of_step'] * 100:.2f} % ({v['GB/s']:.0f} GB/s)" for n, v in gn.items()))
PY
done
cat "$out"
