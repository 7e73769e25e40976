#!/usr/bin/env python3
"""Wall time of each phase of generate_image at BASELINE configs[1] (prints as it goes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", os.environ.get("FM", "FAST"))
import threading
import torch
def _hb():
    while True:
        time.sleep(45); print(f"[{time.perf_counter()-T0:7.1f}s] ...still running", flush=True)
threading.Thread(target=_hb, daemon=True).start()
def log(*a):
    print(f"[{time.perf_counter()-T0:7.1f}s]", *a, flush=True)
T0 = time.perf_counter()
from pigeon_amd import ops, synthetic
B = int(os.environ.get("B", "16")); H = W = int(os.environ.get("RES", "1024"))
dev = "cuda"
pipe = synthetic.build_random_stack(dev, "bf16", seed=123)
torch.cuda.synchronize(); log("stack built; mem GB", torch.cuda.memory_allocated() / 2**30)
tokens = [t.to(dev) for t in synthetic.synthetic_tokens(B)]
def timed(name, fn, n=1):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    log(f"{name}: {dt*1e3:.1f} ms"); return out
with torch.no_grad():
    pe, pooled = timed("embeds (first)", lambda: pipe.generate_image_embeds(tokens))
    timed("embeds", lambda: pipe.generate_image_embeds(tokens), 3)
    lat = torch.randn(B, 4, H // 8, W // 8, device=dev)
    ctx = torch.cat([torch.zeros_like(pe), pe]); add = torch.cat([torch.zeros_like(pooled), pooled])
    ids = torch.tensor([[W, H, 0, 0, W, H]], dtype=torch.bfloat16, device=dev).repeat(2 * B, 1)
    x = ops.cfg_expand_scale(lat, torch.bfloat16, 1.0)
    f = lambda: pipe.unet(x, 961, encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": add, "time_ids": ids}).sample
    timed("unet fwd (first, MIOpen find)", f)
    timed("unet fwd", f, 3)
    torch.cuda.synchronize(); t = time.perf_counter(); f(); t_enq = time.perf_counter() - t; torch.cuda.synchronize()
    log(f"unet fwd host enqueue time (no sync): {t_enq*1e3:.1f} ms")
    if os.environ.get("PROFILE"):
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            f(); torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70), flush=True)
    for chunk in (1, 4, 16):
        if chunk > B: break
        timed(f"vae decode x{chunk} (first)", lambda: pipe.decode_latents(lat[:chunk], "uint8"))
        timed(f"vae decode x{chunk}", lambda: pipe.decode_latents(lat[:chunk], "uint8"), 2)
    log("peak mem GB", torch.cuda.max_memory_allocated() / 2**30)
