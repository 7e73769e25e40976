#!/usr/bin/env python3
"""Three-pass channels-last GroupNorm and the 256 MiB Infinity Cache: apply pass walking the tensor backwards (for
tensors > 200 MB: what the statistics pass read last is what the cache still holds) against forwards.  Measurement
build; interleaved rounds, HIP events, bit-equality checked.  (A third form - sample ranges of <= 96...216 MB processed
back to back - was measured with this script and dropped: profiles/r02_groupnorm_cache_walk_ab.txt.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
lib = use_tuning_lib()
import torch
from pigeon_amd import ops

os.environ["PGN_GN_NHWC_ONEPASS"] = "0"
B = int(os.environ.get("B", "32"))


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


MODES = (("whole, forward", 0, 0), ("whole, reverse (> 200 MB)", 0, 1))
tot = {m[0]: 0.0 for m in MODES}
for C, H, W in ((320, 128, 128), (640, 128, 128), (960, 128, 128), (640, 64, 64), (1280, 64, 64), (1920, 64, 64),
                (1280, 32, 32), (2560, 32, 32)):
    xc = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, device="cuda").bfloat16()
    b = torch.randn(C, device="cuda").bfloat16()
    t = torch.randn(B, C, device="cuda").bfloat16()
    best = {m[0]: 1e9 for m in MODES}
    outs = {}
    for rnd in range(3):
        for name, mb, rev in MODES:
            lib.pgn_groupnorm_nhwc_reverse(rev)
            best[name] = min(best[name], timeit(lambda: ops.group_norm_silu(xc, 32, w, b, 1e-5, True)))
            outs[name] = ops.group_norm_silu(xc, 32, w, b, 1e-5, True, add_nc=t)
    for name in outs:
        assert torch.equal(outs[name], outs[MODES[0][0]]), name
    by = xc.numel() * 4
    print(f"C={C:5d} {H}x{W} ({xc.numel()*2/1e6:6.0f} MB): " + "  ".join(f"{n}: {best[n]*1e6:6.1f} us ({by/best[n]/1e12:.2f})" for n in best), flush=True)
    for k in tot:
        tot[k] += best[k]
print("sum: " + "  ".join(f"{n}: {tot[n]*1e6:.0f} us" for n in tot))
