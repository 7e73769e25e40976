#!/usr/bin/env python3
"""Producer-side GroupNorm statistics at the UNet's shapes (B = UNet batch, default 32 = the bench configuration):
the pair (bias + residual add, GroupNorm + SiLU of its output) as
  plain : pgn_bias_residual_add (in place) + the three GroupNorm launches
  fused : pgn_bias_residual_add_stats (variant v of the measurement build) + finalize + apply
interleaved rounds in one process, HIP events, best of 3 x 20 pairs; also each half of the pair alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
use_tuning_lib()          # pgn_bias_res_stats_variant exists in the measurement build only (make -C pigeon_amd/csrc tuning)
import ctypes
import torch
from pigeon_amd import _lib, ops

B = int(os.environ.get("B", "32"))
SHAPES = ((320, 128, 128), (640, 64, 64), (1280, 32, 32), (640, 128, 128))
VARIANTS = {0: "2 rows in flight, 4 waves/SIMD", 1: "4 rows, 4 waves/SIMD", 2: "2 rows, 8 waves/SIMD (spills)",
            3: "1 row, 8 waves/SIMD"}
lib = _lib.load()
lib.pgn_bias_res_stats_variant.argtypes = [ctypes.c_int]
lib.pgn_bias_res_stats_variant.restype = None


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
    return s.elapsed_time(e) / iters * 1e6


for with_res in (True, False):
    print(f"--- UNet batch {B}, residual {'yes' if with_res else 'no'}")
    for C, H, W in SHAPES:
        mk = lambda: torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        x, res = mk(), (mk() if with_res else None)
        bias, g, b = (torch.randn(C, device="cuda").bfloat16() for _ in range(3))

        def plain_add():
            return ops.bias_residual_add(x, bias, res, inplace=False)

        def fused_add():
            return ops.bias_residual_add(x, bias, res, inplace=False, gn_stats=True)

        y_plain, y_fused = plain_add(), fused_add()
        assert ops.gn_partials(y_fused) is not None and ops.gn_partials(y_plain) is None
        cases = {"plain add": plain_add, "plain GN": lambda: ops.group_norm_silu(y_plain, 32, g, b, 1e-5, True),
                 "plain pair": lambda: ops.group_norm_silu(plain_add(), 32, g, b, 1e-5, True),
                 "pre GN": lambda: ops.group_norm_silu(y_fused, 32, g, b, 1e-5, True)}
        for v in VARIANTS:
            def f_add(v=v):
                lib.pgn_bias_res_stats_variant(v)
                return fused_add()
            cases[f"fused add v{v}"] = f_add
            cases[f"fused pair v{v}"] = lambda v=v: ops.group_norm_silu(cases[f"fused add v{v}"](), 32, g, b, 1e-5, True)
        lib.pgn_groupnorm_nhwc_reverse.argtypes = [ctypes.c_int]
        lib.pgn_groupnorm_nhwc_reverse.restype = None

        def pair_forward_walk():
            lib.pgn_bias_res_stats_variant(0)
            lib.pgn_groupnorm_nhwc_reverse(0)      # apply pass walks the tensor forwards (default: backwards above 200 MB)
            y = ops.group_norm_silu(fused_add(), 32, g, b, 1e-5, True)
            lib.pgn_groupnorm_nhwc_reverse(1)
            return y
        cases["fused pair v0, forward apply walk"] = pair_forward_walk
        best = {k: 1e9 for k in cases}
        for rnd in range(3):
            for k, fn in cases.items():
                best[k] = min(best[k], timeit(fn))
        mb = x.numel() * 2 / 2 ** 20
        print(f"C={C:5d} {H}x{W} ({mb:5.0f} MiB): " + "  ".join(f"{k} {t:6.1f}" for k, t in best.items()), flush=True)
        bp = best["plain pair"]
        print("      pair speed-up: " + "  ".join(f"v{v} x{bp / best[f'fused pair v{v}']:.3f}" for v in VARIANTS), flush=True)
print({v: n for v, n in VARIANTS.items()})
