#!/usr/bin/env python3
"""How much of the self-attention kernel's time depends on WHERE q / k / v rows sit in memory (same bytes, same kernel):
fused-qkv rows at pitch 3C (what the UNet passes), the same with padded rows, separate [B, N, C] tensors, and
head-major contiguous K / V ([B, H, N, 64]: a head's K is one 128 KiB run).  Calls the C ABI directly (strides are
arguments); interleaved rounds in one process."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigeon_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda", 0)
Bp = int(os.environ.get("BP", "32"))


def launch(q, k, v, o, B, H, Nq, Nk, qs, ks, vs, os_):
    rc = lib.pgn_attention_fwd(ops._ptr(q), ops._ptr(k), ops._ptr(v), ops._ptr(o), ops._dt(q), B, H, Nq, Nk, 64,
                               *qs, *ks, *vs, *os_, None, 0, 0.125, ops._stream())
    _lib.check(rc, "pgn_attention_fwd")


def timeit(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


for name, H, N in (("self_L2", 20, 1024), ("self_L1", 10, 4096)):
    C = H * 64
    B = Bp
    fl = 4.0 * B * H * N * N * 64
    out = torch.empty(B, N, C, device=dev, dtype=torch.bfloat16)
    os_ = (N * C, C, 64)
    cases = {}

    def fused(pad):
        P = 3 * C + pad
        buf = torch.randn(B, N, P, device=dev).bfloat16()
        st = (N * P, P, 64)
        q, k, v = buf[:, :, :C], buf[:, :, C:2 * C], buf[:, :, 2 * C:3 * C]
        return lambda: launch(q, k, v, out, B, H, N, N, st, st, st, os_)

    for pad in (0, 64, 128, 192, 256, 512, 1024):
        cases[f"fused qkv, row pitch 3C+{pad}"] = fused(pad)
    q = torch.randn(B, N, C, device=dev).bfloat16(); k = torch.randn(B, N, C, device=dev).bfloat16(); v = torch.randn(B, N, C, device=dev).bfloat16()
    cases["separate [B,N,C] tensors"] = lambda: launch(q, k, v, out, B, H, N, N, os_, os_, os_, os_)
    kh = torch.randn(B, H, N, 64, device=dev).bfloat16(); vh = torch.randn(B, H, N, 64, device=dev).bfloat16()
    hs = (H * N * 64, 64, N * 64)
    cases["q rows [B,N,C], K/V head-major [B,H,N,64]"] = lambda: launch(q, kh, vh, out, B, H, N, N, os_, hs, hs, os_)
    qh = torch.randn(B, H, N, 64, device=dev).bfloat16(); oh = torch.empty(B, H, N, 64, device=dev, dtype=torch.bfloat16)
    cases["all head-major"] = lambda: launch(qh, kh, vh, oh, B, H, N, N, hs, hs, hs, hs)
    res = {c: [] for c in cases}
    for c, fn in cases.items():
        timeit(fn, 3)
    for rnd in range(5):
        for c, fn in cases.items():
            res[c].append(timeit(fn))
    for c in cases:
        m = statistics.median(res[c])
        print(f"{name} {c:48s} {fl / m / 1e12:7.1f} TFLOP/s ({m * 1e6:7.1f} us)", flush=True)
