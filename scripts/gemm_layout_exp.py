"""Experiment: the UNet's GEMM shapes with the weight stored [N, K] (nn.Linear, hipBLASLt 'TN') against a cached
transposed copy [K, N] ('NN'), both through torch with TunableOp tuning on (best solution per shape and layout)."""
import os
import time

os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/gemm_layout_exp.csv")
import torch
import torch.nn.functional as F

dev = "cuda"
SHAPES = [  # (M, K, N, what)
    (32768, 1280, 3840, "qkv L2"), (32768, 1280, 1280, "out L2"), (32768, 1280, 10240, "ff1 L2"), (32768, 5120, 1280, "ff2 L2"),
    (16384, 1280, 1280, "cross q/out L2"),
    (131072, 640, 1920, "qkv L1"), (131072, 640, 640, "out L1"), (131072, 640, 5120, "ff1 L1"), (131072, 2560, 640, "ff2 L1"),
]


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for M, K, N, what in SHAPES:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    wt = w.t().contiguous()
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    flop = 2.0 * M * K * N
    t_tn = timeit(lambda: F.linear(x, w, b))
    t_nn = timeit(lambda: torch.addmm(b, x, wt))
    print(f"{what:16s} M {M:6d} K {K:5d} N {N:5d}: TN {t_tn*1e6:7.1f} us {flop/t_tn/1e12:6.0f} TF | NN {t_nn*1e6:7.1f} us {flop/t_nn/1e12:6.0f} TF  ({t_tn/t_nn:4.2f}x)", flush=True)
