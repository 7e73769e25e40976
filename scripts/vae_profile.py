#!/usr/bin/env python3
"""Kernel breakdown of one VAE decode (1 image, 1024x1024) on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pigeon_amd import synthetic
pipe = synthetic.build_random_stack("cuda", "bf16", seed=123, decoder_depth=1)
lat = torch.randn(int(os.environ.get("B", "1")), 4, 128, 128, device="cuda")
with torch.no_grad():
    for _ in range(2):
        pipe.decode_latents(lat, "uint8")
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        pipe.decode_latents(lat, "uint8"); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=90))
