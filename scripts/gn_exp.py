import sys, os
sys.path.insert(0, os.getcwd())
import torch
from pigeon_amd import ops
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _tuning_lib import use_tuning_lib
lib = use_tuning_lib()
for tw, mr in ((2048, 8), (768, 24), (1024, 32), (512, 24), (768, 32), (768, 16), (1024, 24)):
  lib.pgn_groupnorm_nhwc_tuning(tw, mr)
  tot = 0.0
  for C, HW in ((320, 16384), (640, 16384), (640, 4096), (1280, 4096), (1280, 1024), (1920, 4096), (2560, 1024)):
    side = int(HW ** 0.5)
    xc = torch.randn(32, C, side, side, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, device="cuda").bfloat16(); b = torch.randn(C, device="cuda").bfloat16()
    dt = timeit(lambda: ops.group_norm_silu(xc, 32, w, b, 1e-5, True)); tot += dt
    print(f"  target {tw} minrows {mr}: C={C} HW={HW}: {dt*1e6:7.1f} us {xc.numel()*4/dt/1e9:6.0f} GB/s")
  print(f"target {tw} minrows {mr}: total {tot*1e6:.0f} us")
sys.exit(0)
for C, HW in ((320, 16384), (640, 4096), (1280, 1024), (1920, 4096)):
    side = int(HW ** 0.5)
    x = torch.randn(32, C, side, side, device="cuda").bfloat16()
    xc = x.contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, device="cuda").bfloat16(); b = torch.randn(C, device="cuda").bfloat16()
    for name, t in (("nchw", x), ("nhwc", xc)):
        for act in (True, False):
            dt = timeit(lambda: ops.group_norm_silu(t, 32, w, b, 1e-5, act))
            print(f"{name} C={C} HW={HW} silu={act}: {dt*1e6:7.1f} us  {x.numel()*4/dt/1e9:7.0f} GB/s algorithmic")
