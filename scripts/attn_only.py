#!/usr/bin/env python3
"""A few launches of the attention kernel at one SDXL shape (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigeon_amd import ops, _lib
shape = os.environ.get("SHAPE", "self_L1")
H, Nq, Nk = {"self_L1": (10, 4096, 4096), "self_L2": (20, 1024, 1024), "cross_L1": (10, 4096, 256), "cross_L2": (20, 1024, 256)}[shape]
Bp = 32
if os.environ.get("VAR", "0") != "0":      # 1 = the 8-wave geometry of the measurement build
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _tuning_lib import use_tuning_lib
    use_tuning_lib().pgn_attention_tuning(int(os.environ["VAR"]))
q = torch.randn(Bp, Nq, H * 64, device="cuda").bfloat16(); k = torch.randn(Bp, Nk, H * 64, device="cuda").bfloat16(); v = torch.randn(Bp, Nk, H * 64, device="cuda").bfloat16()
if os.environ.get("LAYOUT", "separate") == "fused" and Nq == Nk:   # q / k / v as column slices of one [B, N, 3C] projection
    C = H * 64
    qkv = torch.randn(Bp, Nq, 3 * C, device="cuda").bfloat16()
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
for _ in range(int(os.environ.get("ITERS", "3"))):
    o = ops.attention(q, k, v, H)
torch.cuda.synchronize()
