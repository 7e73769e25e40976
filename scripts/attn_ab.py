#!/usr/bin/env python3
"""A/B of attention kernel variants of the MEASUREMENT build in one process, interleaved rounds (guide rule 24):
median TFLOP/s per SDXL shape, plus a correctness pass per variant.
usage: attn_ab.py 0 1 3 16     (0 = shipped heuristic, 1 = 4 waves x 32 rows, 3 = 4 waves x 64 rows with LDS-DMA, 16 = 8 waves)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
lib = use_tuning_lib()
import torch
from oracle import ops_ref
from pigeon_amd import ops

variants = [int(v) for v in sys.argv[1:]] or [0]
Bp = int(os.environ.get("BP", "32"))
shapes = (("self_L1", 10, 4096, 4096), ("self_L2", 20, 1024, 1024), ("cross_L1", 10, 4096, 256), ("cross_L2", 20, 1024, 256))
if os.environ.get("SHAPES") == "c2":   # BASELINE configs[2]: 1216x832 images, latent 152x104 (ragged key counts)
    shapes = (("self_L1", 10, 3952, 3952), ("self_L2", 20, 988, 988), ("cross_L1", 10, 3952, 256), ("cross_L2", 20, 988, 256))
weights = {"self_L1": 10 * 2, "self_L2": 60 * 2, "cross_L1": 10, "cross_L2": 60}   # launches per UNet forward (cross: half batch)

def check(var):
    lib.pgn_attention_tuning(var)
    worst = 0.0
    for B, H, Nq, Nk, has_bias in ((2, 10, 1024, 1024, False), (1, 4, 4096, 4096, False), (2, 20, 1024, 256, False), (2, 5, 988, 988, False),
                                   (1, 2, 3952, 3952, False), (2, 4, 300, 77, True), (3, 2, 1, 1, False), (2, 3, 257, 65, False),
                                   (2, 8, 1024, 256, True), (2, 2, 130, 128, False), (2, 2, 64, 129, True), (1, 1, 33, 192, False),
                                   (9, 20, 1024, 1024, False), (40, 16, 256, 1280, False)):
        g = torch.Generator().manual_seed(Nq + Nk)
        C = H * 64
        q = (torch.randn(B, Nq, C, generator=g) * 1.5).bfloat16(); k = (torch.randn(B, Nk, C, generator=g) * 1.5).bfloat16()
        v = torch.randn(B, Nk, C, generator=g).bfloat16()
        if Nk > 200:
            k[:, Nk - 5] = q[:, 0] * 3        # late spike: rescale in a non-first tile
            k[:, 70] = q[:, 1] * 0.6          # moderate early growth: stays under a deferred threshold
        bias = None
        if has_bias:
            keep = torch.randint(1, Nk + 1, (B,), generator=g)
            bias = (1.0 - (torch.arange(Nk)[None, :] < keep[:, None]).float()) * -10000.0
        want = ops_ref.attention_ref(q, k, v, H, None, bias).float()
        got = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, None, None if bias is None else bias.cuda()).cpu().float()
        err = (got - want).abs().max().item() / (2.0 ** -7 * max(1.0, want.abs().max().item()))
        worst = max(worst, err)
        assert not torch.isnan(got).any()
    return worst

for var in variants:
    print(f"variant {var}: worst err / bar = {check(var):.3f}", flush=True)

def timeit(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

res = {}
for name, H, Nq, Nk in shapes:
    C = H * 64
    B = Bp if name.startswith("self") else Bp // 2
    q = torch.randn(B, Nq, C, device="cuda").bfloat16(); k = torch.randn(B, Nk, C, device="cuda").bfloat16(); v = torch.randn(B, Nk, C, device="cuda").bfloat16()
    fl = 4.0 * B * H * Nq * Nk * 64
    for var in variants:
        lib.pgn_attention_tuning(var); timeit(lambda: ops.attention(q, k, v, H), 3)
    for rnd in range(7):
        for var in variants:
            lib.pgn_attention_tuning(var)
            res.setdefault((name, var), []).append(timeit(lambda: ops.attention(q, k, v, H)))
    print(name, "  ".join(f"v{var}: {fl / statistics.median(res[(name, var)]) / 1e12:7.1f} TF ({statistics.median(res[(name, var)]) * 1e6:7.1f} us)" for var in variants), flush=True)
for var in variants:
    tot = sum(weights[n] * statistics.median(res[(n, var)]) for n, *_ in shapes)
    fl = sum(weights[n] * 4.0 * (Bp if n.startswith("self") else Bp // 2) * H * Nq * Nk * 64 for n, H, Nq, Nk in shapes)
    print(f"variant {var}: launch-weighted {fl / tot / 1e12:.1f} TFLOP/s, attention time per UNet forward {tot * 1e3:.2f} ms")
