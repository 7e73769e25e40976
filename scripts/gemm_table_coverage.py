#!/usr/bin/env python3
"""Which GEMMs of one hot-path call are NOT in the shipped TunableOp table (they run hipBLASLt's heuristic pick)?
TunableOp's record-untuned mode with the shipped table loaded and tuning off; one generate_image at the bench shape
(--batch, default 16; one denoising step + VAE decode).  Prints the recorded signatures."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "gemm_untuned.csv"))
args = ap.parse_args()
os.environ["PYTORCH_TUNABLEOP_UNTUNED_FILENAME"] = args.out      # read when the TunableOp context is created
import torch
import torch.cuda.tunable as tn

table = os.path.join(ROOT, "pigeon_amd", "tunableop", "gfx950_bf16.csv")
tn.enable(True)
tn.tuning_enable(False)
tn.set_filename(os.path.join(ROOT, "gpurun_out", "gemm_coverage_scratch.csv"))
assert tn.read_file(table), "the running libraries do not accept the shipped table"
tn.record_untuned_enable(True)
from pigeon_amd import synthetic
pipe = synthetic.build_random_stack(torch.device("cuda", 0), "bf16", seed=123)
toks = [t.cuda() for t in synthetic.synthetic_tokens(args.batch, seed=123)]
pipe.generate_image(image_tokens=toks, width=1024, height=1024, num_inference_steps=1, guidance_scale_for_diffusion=7.0,
                    output_type="uint8")
torch.cuda.synchronize()
stem = os.path.splitext(os.path.basename(args.out))[0]
found = sorted(os.path.join(d, f) for d in {os.path.dirname(args.out), os.getcwd()} for f in os.listdir(d)
               if f.startswith(stem) or f.startswith("tunableop_untuned"))
for f in found:
    rows = open(f).read().strip().splitlines()
    print(f"{f}: {len(rows)} signature(s) outside the table")
    for r in rows:
        print("  ", r)
if not found:
    print("every GEMM of the call is in the table")
