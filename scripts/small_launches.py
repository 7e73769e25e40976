#!/usr/bin/env python3
"""Which host-side torch ops put small kernels into one UNet forward (torch.profiler, shapes recorded): the list the
small-batch launch trimming works from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pigeon_amd import synthetic

B = int(os.environ.get("B", "1"))
pipe = synthetic.build_random_stack("cuda:0", "bf16", seed=123)
tokens = [t.cuda() for t in synthetic.synthetic_tokens(B, seed=123)]
gen = torch.Generator(device="cuda:0").manual_seed(1)
kw = dict(image_tokens=tokens, num_return_images=1, width=1024, height=1024, guidance_scale_for_diffusion=7.0,
          generator=gen, output_type="latent")
pipe.generate_image(num_inference_steps=2, **kw)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    pipe.generate_image(num_inference_steps=4, **kw)
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
rows = [(e.count, e.key, str(e.input_shapes)[:90], e.device_time_total) for e in ev
        if e.key.startswith("aten::") and e.device_time_total > 0 and e.device_time_total / max(e.count, 1) < 30]
rows.sort(key=lambda r: -r[0])
print("count per 4 steps | op | shapes | device us total")
for r in rows[:60]:
    print(f"{r[0]:6d}  {r[1]:32s} {r[2]:92s} {r[3]:9.0f}")
print("small aten launches per step:", sum(r[0] for r in rows) / 4)
