#!/usr/bin/env python3
"""Throughput of the scoring stage at full size: CLIP ViT-H-14 image tower (random init) + pgn_clip_scores on 1024^2 images."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pigeon_amd.clip_vision import VIT_H_14, CLIPScorer, CLIPVisionTransformer
from pigeon_amd.synthetic import _materialize
with torch.device("meta"):
    m = CLIPVisionTransformer(**VIT_H_14)
m = _materialize(m, "cuda", torch.bfloat16, 5)
scorer = CLIPScorer(m, "cuda", torch.bfloat16, batch_size=64)
M = int(os.environ.get("M", "128"))
g = torch.Generator(device="cuda").manual_seed(1)
images = torch.randint(0, 256, (M, 1024, 1024, 3), device="cuda", dtype=torch.uint8, generator=g)
txt = torch.randn(M, 1024, device="cuda"); hist = torch.randn(M, 5, 1024, device="cuda")
for _ in range(2):
    s = scorer.score(images, txt, hist, resize_hw=(512, 512))
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3):
    s = scorer.score(images, txt, hist, resize_hw=(512, 512))
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
px = torch.randn(64, 3, 224, 224, device="cuda").bfloat16()
for _ in range(2): m(px)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): m(px)
torch.cuda.synchronize(); dv = (time.perf_counter() - t) / 5
print(f"scoring stage: {M / dt:.1f} images/s end to end ({dt * 1e3 / M:.2f} ms/image incl. preprocessing); "
      f"ViT-H-14 tower alone {64 / dv:.1f} images/s ({dv * 1e3:.1f} ms per batch of 64); scores finite: {bool(torch.isfinite(s).all())}")
