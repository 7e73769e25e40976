#!/bin/bash
# Wave-cycles / clock / LDS counters of the 32x32x16 ring kernel (variant 7) against its 16x16x32 form (variant 20) at the
# two self-attention shapes: gpurun_out/x16pmc_v<VAR>_<SHAPE>_pmc.json (scripts/attn_pmc.sh per combination).
cd "$(dirname "$0")/.."
export ITERS=20
for SHAPE in self_L1 self_L2; do
  for VAR in 7 20; do
    export SHAPE VAR
    bash scripts/attn_pmc.sh x16pmc_v${VAR}_${SHAPE} || exit 1
  done
done
