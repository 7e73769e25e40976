#!/bin/bash
# Same-box A/B of one host-side restructuring switch (INTEGRATION.md switch table) through bench.py: off / on / off / on.
# usage: scripts/switch_ab.sh <ENV_VARIABLE> <tag>    -> gpurun_out/<tag>.txt
set -u
var=$1
tag=${2:-switch_ab}
out=gpurun_out/$tag.txt
mkdir -p gpurun_out
echo "# scripts/switch_ab.sh $var: bench.py --steps 2 --warmup 1 --no-cpu-baseline, $var = 0 / 1 / 0 / 1 on one box" > "$out"
for sw in 0 1 0 1; do
    env $var=$sw timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag.$sw.json 2> gpurun_out/$tag.$sw.err || exit 1
    python3 - "$var" "$sw" gpurun_out/$tag.$sw.json >> "$out" <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
k = d["kernels"]
print(f"{sys.argv[1]}={sys.argv[2]}: {d['value']:.4f} images/s, {d['ms_per_step']:.1f} ms/step, loop-only "
      f"{d['loop_only']['images_per_sec_per_gpu']:.4f}; " +
      "; ".join(f"{n}: {v['launches']} x {v['avg_us']:.1f} us = {v['share_of_step'] * 100:.2f} %" for n, v in k.items()
                if v['share_of_step'] > 0.002 and not n.startswith('attention')))
PY
done
cat "$out"
