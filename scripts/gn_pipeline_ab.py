#!/usr/bin/env python3
"""Pipelined one-pass channels-last GroupNorm+SiLU (pgn_gn_nhwc_pipe_kernel) against the three launches at the UNet's
shapes (B = UNet batch, default 32 = the bench configuration): interleaved rounds in one process, HIP events, best of
3 x 20 calls; with and without the per-(sample, channel) addend of ResnetBlock2D's norm2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
use_tuning_lib()          # the pipelined form exists in the measurement build only (make -C pigeon_amd/csrc tuning)
import torch
from pigeon_amd import ops

B = int(os.environ.get("B", "32"))
SHAPES = ((320, 128, 128), (640, 128, 128), (960, 128, 128), (640, 64, 64), (1280, 64, 64), (1920, 64, 64),
          (1280, 32, 32), (2560, 32, 32))
FORMS = {"pipe": {"PGN_GN_NHWC_PIPELINE": "1"}, "three": {"PGN_GN_NHWC_PIPELINE": "0", "PGN_GN_NHWC_ONEPASS": "0"}}


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


def use(form):
    for k in ("PGN_GN_NHWC_PIPELINE", "PGN_GN_NHWC_ONEPASS"):
        os.environ.pop(k, None)
    os.environ.update(FORMS[form])


for with_add in (False, True):
    tot = {k: 0.0 for k in FORMS}
    print(f"--- UNet batch {B}, add_nc {'yes' if with_add else 'no'}")
    for C, H, W in SHAPES:
        xc = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        w = torch.randn(C, device="cuda").bfloat16()
        b = torch.randn(C, device="cuda").bfloat16()
        add = torch.randn(B, C, device="cuda").bfloat16() if with_add else None
        best = {k: 1e9 for k in FORMS}
        for rnd in range(3):
            for form in FORMS:
                use(form)
                best[form] = min(best[form], timeit(lambda: ops.group_norm_silu(xc, 32, w, b, 1e-5, True, add_nc=add)))
        by = xc.numel() * 4
        print(f"C={C:5d} {H}x{W} ({xc.numel() * 2 / 2**20:6.0f} MiB): pipelined {best['pipe']*1e6:7.1f} us "
              f"({by/best['pipe']/1e12:.2f} TB/s algorithmic)   three launches {best['three']*1e6:7.1f} us "
              f"({by/best['three']/1e12:.2f})   x{best['three']/best['pipe']:.3f}", flush=True)
        for k in tot:
            tot[k] += best[k]
    print(f"sum over the shapes: pipelined {tot['pipe']*1e6:.0f} us, three launches {tot['three']*1e6:.0f} us, "
          f"x{tot['three']/tot['pipe']:.3f}")
