#!/usr/bin/env python3
"""Rows in flight per lane x register budget of pgn_gn_nhwc_apply_kernel (measurement build: pgn_gn_apply_variant) at
the UNet's batch-32 shapes: the GroupNorm that starts from its producer's statistics (finalize + apply, SiLU) and the
three-launch form with a time-embedding addend (norm2).  Interleaved rounds, best of 3 x 20, us per call."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
use_tuning_lib()
import torch
from pigeon_amd import _lib, ops

B = int(os.environ.get("B", "32"))
SHAPES = ((320, 128, 128), (640, 64, 64), (1280, 32, 32), (640, 128, 128), (1920, 64, 64), (2560, 32, 32))
VARIANTS = {0: "4 rows, 4 waves/SIMD (shipped)", 1: "8 rows, 4 w/SIMD", 2: "2 rows, 8 w/SIMD", 3: "2 rows, 4 w/SIMD",
            4: "4 rows, 8 w/SIMD"}
lib = _lib.load()
lib.pgn_gn_apply_variant.argtypes = [ctypes.c_int]
lib.pgn_gn_apply_variant.restype = None


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


tot = {v: [0.0, 0.0] for v in VARIANTS}
for C, H, W in SHAPES:
    x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    bias, g, b = (torch.randn(C, device="cuda").bfloat16() for _ in range(3))
    add = torch.randn(B, C, device="cuda").bfloat16()
    y = ops.bias_residual_add(x, bias, None, inplace=False, gn_stats=True)
    assert ops.gn_partials(y) is not None
    best = {v: [1e9, 1e9] for v in VARIANTS}
    for rnd in range(3):
        for v in VARIANTS:
            lib.pgn_gn_apply_variant(v)
            best[v][0] = min(best[v][0], timeit(lambda: ops.group_norm_silu(y, 32, g, b, 1e-5, True)))
            best[v][1] = min(best[v][1], timeit(lambda: ops.group_norm_silu(x, 32, g, b, 1e-5, True, add_nc=add)))
    print(f"C={C:5d} {H}x{W}: " + "  ".join(f"v{v}: {t[0]:6.1f} {t[1]:6.1f}" for v, t in best.items()), flush=True)
    for v in VARIANTS:
        tot[v][0] += best[v][0]
        tot[v][1] += best[v][1]
lib.pgn_gn_apply_variant(0)
print("sum (from partials | norm2 three launches): " + "  ".join(f"v{v}: {t[0]:6.1f} {t[1]:6.1f}" for v, t in tot.items()))
print(VARIANTS)
