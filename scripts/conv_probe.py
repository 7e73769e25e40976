#!/usr/bin/env python3
"""3x3 convolution throughput of torch/MIOpen at the SDXL UNet shapes (B' = 32), per memory format and find mode.
Usage: [BENCH=1] [CL=1] python scripts/conv_probe.py     (BENCH -> torch.backends.cudnn.benchmark, CL -> channels_last)"""
import os, sys, time
import torch
import torch.nn.functional as F

bench = bool(os.environ.get("BENCH"))
cl = bool(os.environ.get("CL"))
torch.backends.cudnn.benchmark = bench
dev = "cuda"
Bp = int(os.environ.get("BP", "32"))
shapes = [(320, 320, 128), (640, 320, 128), (960, 320, 128), (320, 640, 64), (640, 640, 64), (1280, 640, 64),
          (1920, 640, 64), (960, 640, 64), (640, 1280, 32), (1280, 1280, 32), (2560, 1280, 32), (1920, 1280, 32)]
print(f"benchmark={bench} channels_last={cl} FIND_MODE={os.environ.get('MIOPEN_FIND_MODE')}", flush=True)
tot = 0.0
for cin, cout, hw in shapes:
    x = torch.randn(Bp, cin, hw, hw, device=dev).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=dev).bfloat16() * 0.02
    b = torch.zeros(cout, device=dev).bfloat16()
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
        w = w.contiguous(memory_format=torch.channels_last)
    for _ in range(2):
        y = F.conv2d(x, w, b, padding=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        y = F.conv2d(x, w, b, padding=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 2.0 * Bp * hw * hw * cin * cout * 9
    tot += dt
    print(f"conv {cin:5d}->{cout:5d} @{hw:3d}^2  {dt*1e6:9.1f} us  {fl/dt/1e12:7.1f} TFLOP/s  out_cl={y.is_contiguous(memory_format=torch.channels_last)}", flush=True)
print(f"sum {tot*1e3:.2f} ms")
