#!/usr/bin/env python3
"""Which torch (non-GEMM, non-conv, non-pgn) ops a UNet forward at BASELINE configs[1] still launches, with shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pigeon_amd import ops, synthetic
B, H, W, dev = 16, 1024, 1024, "cuda"
pipe = synthetic.build_random_stack(dev, "bf16", seed=123, with_vae=False, decoder_depth=1)
with torch.no_grad():
    lat = torch.randn(B, 4, H // 8, W // 8, device=dev)
    ctx = torch.randn(2 * B, 256, 2048, device=dev).bfloat16(); add = torch.randn(2 * B, 1280, device=dev).bfloat16()
    ids = torch.tensor([[W, H, 0, 0, W, H]], dtype=torch.bfloat16, device=dev).repeat(2 * B, 1)
    x = ops.cfg_expand_scale(lat, torch.bfloat16, 1.0)
    f = lambda: pipe.unet(x, 961, encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": add, "time_ids": ids}).sample
    f(); f(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        f(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if t > 0:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total device us", tot)
for t, n, k, s in rows[:60]:
    print(f"{t:10.0f} us {100*t/tot:5.1f}%  x{n:4d}  {k:40s} {s}")
