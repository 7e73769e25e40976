"""How the 64-row attention kernel's throughput depends on the number of workgroup rounds (workgroups / 512 resident
slots) and on the key length: separates per-workgroup overhead (prologue / epilogue) from tail effects."""
import torch

from pigeon_amd import ops

dev = "cuda"


def run(B, H, N, reps=20):
    qkv = torch.randn(B, N, 3 * H * 64, device=dev, dtype=torch.bfloat16)
    q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
    for _ in range(3):
        ops.attention(q, k, v, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.attention(q, k, v, H)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flop = 4.0 * B * H * N * N * 64
    wgs = B * H * ((N + 255) // 256)
    print(f"B {B:4d} H {H:3d} N {N:5d}: {wgs:6d} workgroups = {wgs/512:6.2f} rounds, {us:8.1f} us, {flop/us/1e6:7.1f} TFLOP/s,"
          f" {us/ (wgs/512) / (N/64):6.3f} us per round per key tile", flush=True)


for N in (1024, 2048, 4096):
    for B in (8, 16, 32, 64, 128):
        if B * N > 128 * 2048 and N > 2048:
            continue
        run(B, 16, N)
run(32, 20, 1024)
run(32, 10, 4096)
