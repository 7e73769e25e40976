#!/usr/bin/env python3
"""Where a ring-kernel workgroup's life goes (measurement build): per-workgroup s_memrealtime stamps at entry, after the
prologue barrier (Q fragments + tile 0 landed), after the key loop and after the output stores have left, plus the gap
between one workgroup's end and the start of the next one in the same hardware slot (XCC, SE, CU, wave slot)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
lib = use_tuning_lib()
import ctypes
import torch
from pigeon_amd import ops

lib.pgn_attention_debug_buffer.argtypes = [ctypes.c_void_p]
lib.pgn_attention_debug_buffer.restype = None
lib.pgn_attention_tuning(7)
Bp = int(os.environ.get("BP", "32"))
for name, H, N in (("self_L2", 20, 1024), ("self_L1", 10, 4096)):
    C = H * 64
    qkv = torch.randn(Bp, N, 3 * C, device="cuda").bfloat16()
    q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    nwg = Bp * H * (N // 256)
    dbg = torch.zeros(nwg * 6, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.attention(q, k, v, H)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for _ in range(10):
        ops.attention(q, k, v, H)
    ev[1].record()
    lib.pgn_attention_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    ev[2].record()
    for _ in range(10):
        ops.attention(q, k, v, H)
    ev[3].record()
    torch.cuda.synchronize()
    print(f"{name}: {ev[0].elapsed_time(ev[1]) * 100:.1f} us per launch without stamps, {ev[2].elapsed_time(ev[3]) * 100:.1f} us with")
    lib.pgn_attention_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    ops.attention(q, k, v, H)
    torch.cuda.synchronize()
    lib.pgn_attention_debug_buffer(None)
    d = dbg.cpu().view(nwg, 6).numpy()
    tick = 0.01  # us per s_memrealtime tick (100 MHz)
    t0 = d[:, 0].min()
    pro = (d[:, 1] - d[:, 0]) * tick
    loop = (d[:, 2] - d[:, 1]) * tick
    epi = (d[:, 3] - d[:, 2]) * tick
    life = (d[:, 3] - d[:, 0]) * tick
    nt = N // 64
    print(f"{name}: {nwg} workgroups, {nt} key tiles each; kernel span {(d[:, 3].max() - t0) * tick:.1f} us")
    for lbl, x in (("prologue (entry -> Q + tile 0 landed, barrier)", pro), ("key loop", loop), ("  per tile", loop / nt),
                   ("epilogue (normalise, stage, store, stores retired)", epi), ("whole life", life)):
        print(f"   {lbl:52s} median {statistics.median(x):7.2f} us   p10 {sorted(x)[len(x) // 10]:7.2f}   p90 {sorted(x)[len(x) * 9 // 10]:7.2f}")
    # how many workgroups are alive over the kernel's span (512 = every slot of the chip)
    import numpy as np
    span = float(d[:, 3].max() - t0)
    alive = float((d[:, 3] - d[:, 0]).sum()) / span
    print(f"   workgroups alive on average over the span: {alive:.1f} of 512 slots")
    starts = np.sort(d[:, 0] - t0) * tick
    print(f"   entry time of the 128th / 256th / 384th / 512th workgroup: {starts[127]:.2f} / {starts[255]:.2f} / {starts[383]:.2f} / {starts[511]:.2f} us")
    # slot = (xcc, se, sh, cu of HW_ID): consecutive workgroups of one CU
    slots = {}
    for i in range(nwg):
        key = (int(d[i, 5]) & 15, int(d[i, 4]) & 0xff00)
        slots.setdefault(key, []).append((int(d[i, 0]), int(d[i, 3])))
    gaps = []
    first = []
    for key, lst in slots.items():
        lst.sort()
        first.append((lst[0][0] - t0) * tick)
        # two workgroups share a CU: the k-th entry follows the (k-2)-th exit
        ends_sorted = sorted(e for _, e in lst)
        for k in range(2, len(lst)):
            gaps.append((lst[k][0] - ends_sorted[k - 2]) * tick)
    print(f"   {len(slots)} CUs seen; first workgroup of a CU starts {statistics.median(first):.2f} us after the first of the launch (p90 {sorted(first)[len(first) * 9 // 10]:.2f})")
    if gaps:
        print(f"   gap between the k-th exit and the (k+2)-th entry on one CU (slot turnaround): median {statistics.median(gaps):.2f} us, p10 {sorted(gaps)[len(gaps) // 10]:.2f}, p90 {sorted(gaps)[len(gaps) * 9 // 10]:.2f}")
    ends = sorted((int(x) - t0) * tick for x in d[:, 3])
    print(f"   last workgroup ends at {ends[-1]:.1f} us; 90 % of the workgroups have ended by {ends[len(ends) * 9 // 10]:.1f} us")
