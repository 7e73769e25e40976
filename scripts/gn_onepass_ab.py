#!/usr/bin/env python3
"""One-pass against three-pass channels-last GroupNorm+SiLU at the UNet's batch-32 shapes (shipped library, the form is
switched with PGN_GN_NHWC_ONEPASS, read on every call): interleaved rounds in one process, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigeon_amd import ops

B = int(os.environ.get("B", "32"))
SHAPES = ((320, 128, 128), (640, 128, 128), (960, 128, 128), (640, 64, 64), (1280, 64, 64), (1920, 64, 64),
          (1280, 32, 32), (2560, 32, 32))


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


tot = {"1": 0.0, "0": 0.0}
for C, H, W in SHAPES:
    xc = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, device="cuda").bfloat16()
    b = torch.randn(C, device="cuda").bfloat16()
    best = {"1": 1e9, "0": 1e9}
    for rnd in range(3):
        for form in ("1", "0"):
            os.environ["PGN_GN_NHWC_ONEPASS"] = form
            best[form] = min(best[form], timeit(lambda: ops.group_norm_silu(xc, 32, w, b, 1e-5, True)))
    by = xc.numel() * 4
    print(f"C={C:5d} {H}x{W}: one-pass {best['1']*1e6:7.1f} us ({by/best['1']/1e12:.2f} TB/s algorithmic)   "
          f"three-pass {best['0']*1e6:7.1f} us ({by/best['0']/1e12:.2f})   x{best['0']/best['1']:.3f}", flush=True)
    for k in tot:
        tot[k] += best[k]
print(f"sum over the shapes: one-pass {tot['1']*1e6:.0f} us, three-pass {tot['0']*1e6:.0f} us, x{tot['0']/tot['1']:.3f}")
