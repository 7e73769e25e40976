"""Experiment: 3x3 convolutions of the SDXL UNet as ONE GEMM over an explicit channels-last im2col matrix
([N*H*W, 9*Cin] x [Cout, 9*Cin]^T, K = 9*Cin is large: hipBLASLt's best regime) against MIOpen's tuned CK
implicit-GEMM convolution.  Prints time and TFLOP/s of both plus the HBM time an im2col pass would add."""
import sys
import time

import torch
import torch.nn.functional as F

import pigeon_amd  # noqa: F401  (points MIOpen at the shipped find-db)

dev = "cuda"
SHAPES = [  # (Cin, Cout, H=W, batch)
    (1280, 1280, 32, 32), (2560, 1280, 32, 32), (1920, 1280, 32, 32), (640, 1280, 32, 32), (1280, 1280, 64, 32),
    (640, 640, 64, 32), (1280, 640, 64, 32), (1920, 640, 64, 32), (960, 640, 64, 32), (640, 640, 128, 32),
    (320, 320, 128, 32), (960, 320, 128, 32), (640, 320, 128, 32),
]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for Cin, Cout, H, B in SHAPES:
    x = torch.randn(B, Cin, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev, dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev, dtype=torch.bfloat16)
    M, K = B * H * H, 9 * Cin
    flop = 2.0 * M * K * Cout
    t_conv = timeit(lambda: F.conv2d(x, w, None, padding=1))
    if M * K * 2 > 40e9:
        print(f"{Cin:5d}->{Cout:5d} @{H:3d}: conv {t_conv*1e3:7.3f} ms {flop/t_conv/1e12:6.0f} TF | im2col matrix too large, skipped")
        continue
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W2 = w.permute(0, 2, 3, 1).reshape(Cout, K).contiguous()
    t_gemm = timeit(lambda: F.linear(A, W2, b))
    t_im2col = (M * K * 2 + M * Cin * 2) / 5.0e12          # write the matrix + read the activation at 5 TB/s
    tot = t_gemm + t_im2col
    print(f"{Cin:5d}->{Cout:5d} @{H:3d}: conv {t_conv*1e3:7.3f} ms {flop/t_conv/1e12:6.0f} TF | gemm {t_gemm*1e3:7.3f} ms "
          f"{flop/t_gemm/1e12:6.0f} TF + im2col {t_im2col*1e3:6.3f} ms = {tot*1e3:7.3f} ms ({t_conv/tot:4.2f}x)", flush=True)
    del A
