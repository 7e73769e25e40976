#!/usr/bin/env python3
"""LayerNorm launch times at the UNet's shapes: plain, with the pending column bias, with residual + column bias
(shipped library; HIP events, best of 3 x 30 calls).  The plain form is the same-box reference for the other two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigeon_amd import ops


def timeit(fn, iters=30, warm=3):
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


for M, C in ((131072, 640), (32768, 1280), (16384, 1280), (65536, 640), (8192, 1280), (2048, 1280)):
    x = torch.randn(M, C, device="cuda").bfloat16()
    res = torch.randn(M, C, device="cuda").bfloat16()
    w = torch.randn(C, device="cuda").bfloat16()
    b = torch.randn(C, device="cuda").bfloat16()
    cb = torch.randn(C, device="cuda")
    out = []
    for tag, residual, colb in (("plain", None, None), ("col_bias", None, cb), ("residual+col_bias", res, cb)):
        t = min(timeit(lambda: ops.add_layer_norm(x, residual, w, b, 1e-5, col_bias=colb)) for _ in range(3))
        by = x.numel() * 2 * (4 if residual is not None else 2)
        out.append(f"{tag}: {t*1e6:6.1f} us ({by/t/1e12:.2f} TB/s)")
    print(f"M={M} C={C}  " + "   ".join(out), flush=True)
