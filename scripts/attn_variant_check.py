#!/usr/bin/env python3
"""Correctness of an attention kernel variant (pgn_attention_tuning) against the CPU oracle on the test shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ops_ref
from pigeon_amd import ops, _lib
var = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
use_tuning_lib().pgn_attention_tuning(var)
bad = 0
for B, H, Nq, Nk, has_bias in ((2, 10, 1024, 1024, False), (1, 4, 4096, 4096, False), (2, 20, 1024, 256, False), (2, 5, 988, 988, False),
                               (1, 2, 3952, 3952, False), (2, 4, 300, 77, True), (3, 2, 1, 1, False), (2, 3, 257, 65, False),
                               (2, 8, 1024, 256, True), (2, 2, 130, 128, False), (2, 2, 64, 129, True), (1, 1, 33, 192, False)):
    g = torch.Generator().manual_seed(Nq + Nk)
    C = H * 64
    q = (torch.randn(B, Nq, C, generator=g) * 1.5).bfloat16(); k = (torch.randn(B, Nk, C, generator=g) * 1.5).bfloat16()
    v = torch.randn(B, Nk, C, generator=g).bfloat16()
    if Nk > 200:  # late spike -> rescale branch in a non-first tile
        k[:, Nk - 5] = q[:, 0] * 3
    bias = None
    if has_bias:
        keep = torch.randint(1, Nk + 1, (B,), generator=g)
        bias = (1.0 - (torch.arange(Nk)[None, :] < keep[:, None]).float()) * -10000.0
    want = ops_ref.attention_ref(q, k, v, H, None, bias).float()
    got = ops.attention(q.cuda(), k.cuda(), v.cuda(), H, None, None if bias is None else bias.cuda()).cpu().float()
    err = (got - want).abs().max().item(); tol = 2.0 ** -7 * max(1.0, want.abs().max().item())
    ok = err < tol and not torch.isnan(got).any()
    bad += (not ok)
    print(f"B{B} H{H} q{Nq} k{Nk} bias={has_bias}: err {err:.5f} tol {tol:.5f} {'ok' if ok else 'FAIL'}", flush=True)
sys.exit(1 if bad else 0)
