"""What the feed-forward's first GEMM (+ GEGLU) costs under the package's own GEMM table: the number a fused
GEMM + GEGLU kernel has to beat."""
import time

import torch
import torch.nn.functional as F

import pigeon_amd  # noqa: F401  (TunableOp table in look-up mode, as the pipeline runs)
from pigeon_amd import ops

dev = "cuda"


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for M, C in ((32768, 1280), (131072, 640)):
    x = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
    w = torch.randn(8 * C, C, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.randn(8 * C, device=dev, dtype=torch.bfloat16)
    w2 = torch.randn(C, 4 * C, device=dev, dtype=torch.bfloat16) * 0.02
    res = torch.randn(M, C, device=dev, dtype=torch.bfloat16)
    h = F.linear(x, w, b)
    t1 = timeit(lambda: F.linear(x, w, b))
    t2 = timeit(lambda: ops.geglu(h))
    g = ops.geglu(h)
    t3 = timeit(lambda: res.addmm_(g, w2.t()))
    f1, f3 = 2.0 * M * C * 8 * C, 2.0 * M * 4 * C * C
    print(f"M {M} C {C}: ff1 {t1*1e6:7.1f} us ({f1/t1/1e12:5.0f} TF)  geglu {t2*1e6:6.1f} us  ff2(beta=1) {t3*1e6:7.1f} us ({f3/t3/1e12:5.0f} TF)"
          f"  | fused ff1+geglu break-even {f1/(t1+t2)/1e12:5.0f} TF", flush=True)
