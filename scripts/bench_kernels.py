#!/usr/bin/env python3
"""Micro-benchmarks of the HIP kernels at the BASELINE config-2 shapes (B' = 32 = 2 x batch 16, 128x128 latent).
Prints achieved TFLOP/s (attention) and GB/s (GroupNorm, step) from HIP-event timing on torch's current stream."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigeon_amd import ops

dev = "cuda"
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

res = []
Bp = int(os.environ.get("BP", "32"))
shapes_cfg5 = (("self_L1_2x", 10, 16384, 16384), ("self_L2_2x", 20, 4096, 4096), ("cross_L1_2x_k1024", 10, 16384, 1024),
               ("cross_L2_2x_k1024", 20, 4096, 1024))   # BASELINE configs[4]: 2x latent side, long cross-attention
for name, H, Nq, Nk in shapes_cfg5 if os.environ.get("SHAPES") == "cfg5" else (("self_L1", 10, 4096, 4096), ("self_L2", 20, 1024, 1024), ("cross_L1", 10, 4096, 256), ("cross_L2", 20, 1024, 256),
                        ("self_L1_ragged", 10, 3952, 3952), ("self_L2_ragged", 20, 988, 988)):
    C = H * 64
    q = torch.randn(Bp, Nq, C, device=dev).bfloat16(); k = torch.randn(Bp, Nk, C, device=dev).bfloat16(); v = torch.randn(Bp, Nk, C, device=dev).bfloat16()
    fl = 4.0 * Bp * H * Nq * Nk * 64
    from pigeon_amd import _lib
    variants = ((0, "4w"),)
    if os.environ.get("TUNING"):       # measurement build (make -C pigeon_amd/csrc tuning): also the 8-wave geometry
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _tuning_lib import use_tuning_lib
        use_tuning_lib()
        variants = ((1, "8w"), (0, "4w"))
    for var, vn in variants:
        if os.environ.get("TUNING"):
            _lib.load().pgn_attention_tuning(var)
        t = timeit(lambda: ops.attention(q, k, v, H))
        res.append((f"attn_{name}_{vn}", t * 1e6, fl / t / 1e12, "TFLOP/s"))
    tq = q.view(Bp, Nq, H, 64).transpose(1, 2); tk = k.view(Bp, Nk, H, 64).transpose(1, 2); tv = v.view(Bp, Nk, H, 64).transpose(1, 2)
    if os.environ.get("SDPA"):
        t2 = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(tq, tk, tv), iters=5, warm=2)
        res.append((f"  torch_sdpa_{name}", t2 * 1e6, fl / t2 / 1e12, "TFLOP/s"))
if os.environ.get("SHAPES") == "cfg5":
    for r in res:
        print(f"{r[0]:34s} {r[1]:10.1f} us  {r[2]:9.1f} {r[3]}")
    sys.exit(0)
for C, HW in ((320, 16384), (640, 16384), (960, 16384), (640, 4096), (1280, 4096), (1920, 4096), (1280, 1024), (2560, 1024)):
    x = torch.randn(Bp, C, HW, device=dev).bfloat16(); w = torch.randn(C, device=dev).bfloat16(); b = torch.randn(C, device=dev).bfloat16()
    t = timeit(lambda: ops.group_norm_silu(x, 32, w, b, 1e-5, True))
    res.append((f"gn_silu_{C}x{HW}", t * 1e6, x.numel() * 4 / t / 1e9, "GB/s"))
    xc = x.view(Bp, C, int(HW ** 0.5), -1).contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: ops.group_norm_silu(xc, 32, w, b, 1e-5, True))
    res.append((f"gn_silu_nhwc_{C}x{HW}", t * 1e6, x.numel() * 4 / t / 1e9, "GB/s"))
    if os.environ.get("TORCHGN"):
        gn = torch.nn.GroupNorm(32, C).to(dev).bfloat16()
        t2 = timeit(lambda: torch.nn.functional.silu(gn(x)), iters=5, warm=2)
        res.append((f"  torch_gn_silu_{C}x{HW}", t2 * 1e6, x.numel() * 4 / t2 / 1e9, "GB/s"))
for C, HW in ((320, 16384), (640, 4096), (1280, 1024)):
    side = int(HW ** 0.5)
    x = torch.randn(Bp, C, side, side, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    r = torch.randn_like(x); b = torch.randn(C, device=dev).bfloat16()
    t = timeit(lambda: ops.bias_residual_add(x, b, r))
    res.append((f"bias_residual_nhwc_{C}x{HW}", t * 1e6, x.numel() * 6 / t / 1e9, "GB/s"))
    xn = x.contiguous(); rn = r.contiguous()
    t = timeit(lambda: ops.bias_residual_add(xn, b, rn))
    res.append((f"bias_residual_nchw_{C}x{HW}", t * 1e6, x.numel() * 6 / t / 1e9, "GB/s"))
for B in (16, 128):
    npred = torch.randn(2 * B, 4, 128, 128, device=dev).bfloat16(); lat = torch.randn(B, 4, 128, 128, device=dev)
    t = timeit(lambda: ops.cfg_ddim_step(npred, lat, 7.0, 0.9, 0.43, 0.92, 0.39))
    res.append((f"cfg_ddim_step_B{B}", t * 1e6, lat.numel() * 12 / t / 1e9, "GB/s"))
for r in res:
    print(f"{r[0]:34s} {r[1]:10.1f} us  {r[2]:9.1f} {r[3]}")
