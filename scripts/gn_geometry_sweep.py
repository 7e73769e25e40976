#!/usr/bin/env python3
"""Workgroup geometry of the channels-last GroupNorm launches (target workgroups per launch, minimum rows per lane) with
the producer-side statistics in place: per geometry, the (bias + residual add with statistics -> finalize -> apply)
pair and the three-launch form with a time-embedding addend (norm2), at the UNet's batch-32 shapes.  Measurement build
(pgn_groupnorm_nhwc_tuning); interleaved rounds, best of 3 x 20."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
use_tuning_lib()
import torch
from pigeon_amd import _lib, ops

B = int(os.environ.get("B", "32"))
SHAPES = ((320, 128, 128), (640, 64, 64), (1280, 32, 32), (640, 128, 128), (1920, 64, 64))
GEOMS = [(768, 24), (512, 24), (1024, 24), (1536, 24), (768, 16), (1024, 16), (1536, 16), (2048, 16), (1536, 8), (3072, 8)]
lib = _lib.load()
lib.pgn_groupnorm_nhwc_tuning.argtypes = [ctypes.c_int, ctypes.c_int]
lib.pgn_groupnorm_nhwc_tuning.restype = None


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
    return s.elapsed_time(e) / iters * 1e3


tot = {g: [0.0, 0.0] for g in GEOMS}
for C, H, W in SHAPES:
    mk = lambda: torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    x, res = mk(), mk()
    bias, g, b = (torch.randn(C, device="cuda").bfloat16() for _ in range(3))
    add = torch.randn(B, C, device="cuda").bfloat16()
    best = {gm: [1e9, 1e9] for gm in GEOMS}
    for rnd in range(3):
        for gm in GEOMS:
            lib.pgn_groupnorm_nhwc_tuning(*gm)
            pair = lambda: ops.group_norm_silu(ops.bias_residual_add(x, bias, res, inplace=False, gn_stats=True), 32, g, b, 1e-5, True)
            three = lambda: ops.group_norm_silu(x, 32, g, b, 1e-5, True, add_nc=add)
            best[gm][0] = min(best[gm][0], timeit(pair))
            best[gm][1] = min(best[gm][1], timeit(three))
    print(f"C={C:5d} {H}x{W}: " + "  ".join(f"{gm[0]}/{gm[1]}: {v[0]:6.1f} {v[1]:6.1f}" for gm, v in best.items()), flush=True)
    for gm in GEOMS:
        tot[gm][0] += best[gm][0]
        tot[gm][1] += best[gm][1]
print("sum (pair us, norm2 three-launch us): " + "  ".join(f"{gm[0]}/{gm[1]}: {v[0]:6.1f} {v[1]:6.1f}" for gm, v in tot.items()))
lib.pgn_groupnorm_nhwc_tuning(768, 24)
