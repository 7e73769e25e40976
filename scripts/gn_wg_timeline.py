#!/usr/bin/env python3
"""Where a one-pass GroupNorm workgroup's life goes (measurement build): s_memrealtime stamps at entry, rows landed and
summed, words published, wait over, statistics ready, stores issued, stores retired."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
lib = use_tuning_lib()
import ctypes
import numpy as np
import torch
from pigeon_amd import ops

lib.pgn_groupnorm_nhwc_debug_buffer.argtypes = [ctypes.c_void_p]
lib.pgn_groupnorm_nhwc_debug_buffer.restype = None
B = int(os.environ.get("B", "32"))
for C, H, W in ((320, 128, 128), (640, 64, 64), (1280, 32, 32)):
    xc = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, device="cuda").bfloat16()
    b = torch.randn(C, device="cuda").bfloat16()
    for _ in range(3):
        ops.group_norm_silu(xc, 32, w, b, 1e-5, True)
    dbg = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.pgn_groupnorm_nhwc_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    ops.group_norm_silu(xc, 32, w, b, 1e-5, True)
    torch.cuda.synchronize()
    lib.pgn_groupnorm_nhwc_debug_buffer(None)
    d = dbg.cpu().view(-1, 8).numpy()
    nwg = int((d[:, 0] != 0).sum())
    d = d[:nwg]
    tick = 0.01
    t0 = d[:, 0].min()
    print(f"C={C} {H}x{W} batch {B}: {nwg} workgroups ({nwg // B} per sample); launch span {(d[:, 6].max() - t0) * tick:.1f} us")
    names = ("entry -> rows landed and summed", "fold + publish (drained, barrier)", "arrive + wait for the sample",
             "acquire + S x G reduction", "apply + stores issued", "stores retired", "whole life")
    cols = [d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 4] - d[:, 3], d[:, 5] - d[:, 4], d[:, 6] - d[:, 5],
            d[:, 6] - d[:, 0]]
    for nm, x in zip(names, cols):
        x = np.sort(x * tick)
        print(f"   {nm:36s} median {x[len(x) // 2]:7.2f} us   p10 {x[len(x) // 10]:7.2f}   p90 {x[len(x) * 9 // 10]:7.2f}")
    alive = float((d[:, 6] - d[:, 0]).sum()) / float(d[:, 6].max() - t0)
    print(f"   workgroups alive on average over the span: {alive:.1f}")
    starts = np.sort(d[:, 0] - t0) * tick
    for k in (255, 511, 767, 1023):
        if k < nwg:
            print(f"   entry of workgroup #{k + 1} (by time): {starts[k]:.2f} us")
