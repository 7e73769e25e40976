"""Is the N = 1024 deficit of the 64-row attention kernel (2.5 vs 2.1 us per key tile) K / V fetch latency?  Same launch
with every (batch, head) reading ONE shared K / V (strides 0: always L2 hits) against the normal layout."""
import torch

from pigeon_amd import _lib

lib = _lib.load()
dev = "cuda"


def run(B, H, N, shared, reps=20):
    C = H * 64
    q = torch.randn(B, N, C, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, N, C, device=dev, dtype=torch.bfloat16)
    v = torch.randn(B, N, C, device=dev, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    ksb, ksh = (0, 0) if shared else (N * C, 64)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        rc = lib.pgn_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 1, B, H, N, N, 64,
                                   N * C, C, 64, ksb, C, ksh, ksb, C, ksh, N * C, C, 64, None, 0, 0.125, st)
        assert rc == 0, rc
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flop = 4.0 * B * H * N * N * 64
    print(f"B {B} H {H} N {N} shared K/V {shared}: {us:8.1f} us {flop/us/1e6:7.1f} TFLOP/s", flush=True)


for N, B, H in ((1024, 32, 20), (4096, 32, 10), (2048, 32, 16)):
    run(B, H, N, False)
    run(B, H, N, True)
