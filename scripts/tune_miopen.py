#!/usr/bin/env python3
"""Fill a MIOpen user find-db for the convolutions of the hot path (run on the MI355X box).

MIOpen's immediate-mode heuristic picks its `igemm_fwd_gtcx35_nhwc` solver for the SDXL 3x3 convolutions at
300-500 TFLOP/s; a find (torch.backends.cudnn.benchmark) measures every applicable solver and reaches 500-900.
A find costs ~15 s per convolution shape, so it is done once, here, and the resulting text database
(`*.ufdb.txt`) is shipped in-tree under pigeon_amd/miopen_db/ (pigeon_amd/__init__.py points MIOPEN_USER_DB_PATH at it).

  MIOPEN_USER_DB_PATH=gpurun_out/miopen_db python scripts/tune_miopen.py [--batch 16] [--height 1024] [--width 1024]

Step 1 records every distinct (input shape, weight shape, stride, padding, dtype, memory format) of one UNet forward at
B' = 2 x batch and one VAE decode at batch 1 (find off); step 2 runs each once with find on, printing as it goes."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--shapes", default="gpurun_out/conv_shapes.json")
    ap.add_argument("--no-find", action="store_true", help="step 2 with cudnn.benchmark off (checks that immediate "
                    "mode picks the shipped find-db entries)")
    ap.add_argument("--collect", action="store_true", help="step 1 only (own process: torch caches the immediate-mode "
                    "choice per shape, so a find in the same process would never run)")
    args = ap.parse_args()
    if args.collect:
        return collect(args)
    return tune(args)


def collect(args):
    from pigeon_amd import synthetic
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = False
    pipe = synthetic.build_random_stack(dev, "bf16", seed=123)
    seen = {}

    # every convolution of the stack ends in torch.nn.functional.conv2d (nn.Conv2d.forward and the bias-less calls of
    # pigeon_amd.layers alike): record there
    orig_conv2d = F.conv2d

    def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
        key = (tuple(x.shape), tuple(weight.shape), pair(stride), pair(padding), str(x.dtype)[6:],
               x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous(), False)
        seen.setdefault(key, 0)
        seen[key] += 1
        return orig_conv2d(x, weight, bias, stride, padding, dilation, groups)

    F.conv2d = conv2d
    torch.nn.functional.conv2d = conv2d
    h, w = args.height // 8, args.width // 8
    bp = 2 * args.batch
    with torch.no_grad():
        x = torch.randn(bp, 4, h, w, device=dev).bfloat16()
        ctx = torch.randn(bp, 256, 2048, device=dev).bfloat16()
        added = {"text_embeds": torch.randn(bp, 1280, device=dev).bfloat16(),
                 "time_ids": torch.tensor([[args.width, args.height, 0, 0, args.width, args.height]], device=dev).repeat(bp, 1)}
        pipe.unet(x, 981, encoder_hidden_states=ctx, added_cond_kwargs=added)
        if not args.no_vae:
            pipe.vae.decode(torch.randn(1, 4, h, w, device=dev).bfloat16())
    torch.cuda.synchronize()
    import json
    with open(args.shapes, "w") as fh:
        json.dump([[list(k), c] for k, c in seen.items()], fh)
    print(f"{len(seen)} distinct convolutions -> {args.shapes}", flush=True)


def tune(args):
    import json
    assert os.environ.get("MIOPEN_USER_DB_PATH"), "set MIOPEN_USER_DB_PATH to the directory that collects the db"
    os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
    dev = torch.device("cuda", 0)
    with open(args.shapes) as fh:
        seen = {(tuple(k[0]), tuple(k[1]), tuple(k[2]), tuple(k[3]), getattr(torch, k[4]), k[5], k[6]): c
                for k, c in json.load(fh)}
    torch.backends.cudnn.benchmark = not args.no_find
    t_all = time.perf_counter()
    for i, (key, cnt) in enumerate(sorted(seen.items(), key=lambda kv: str(kv[0]))):
        xs, ws, stride, pad, dt, cl, has_bias = key
        x = torch.randn(xs, device=dev).to(dt)
        wgt = (torch.randn(ws, device=dev) * 0.02).to(dt)
        b = torch.zeros(ws[0], device=dev).to(dt) if has_bias else None
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
            wgt = wgt.contiguous(memory_format=torch.channels_last)
        t0 = time.perf_counter()
        y = F.conv2d(x, wgt, b, stride=stride, padding=pad)
        torch.cuda.synchronize()
        t_find = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(3):
            y = F.conv2d(x, wgt, b, stride=stride, padding=pad)
        torch.cuda.synchronize()
        dt_run = (time.perf_counter() - t0) / 3
        fl = 2.0 * y.numel() * ws[1] * ws[2] * ws[3]
        print(f"[{i + 1}/{len(seen)}] x{cnt} in={xs} w={ws} s={stride} p={pad} cl={cl}: find {t_find:.1f} s, "
              f"{dt_run * 1e6:.0f} us, {fl / dt_run / 1e12:.0f} TFLOP/s", flush=True)
        del x, wgt, y
    # manifest of tuned (UNet batch, latent h, latent w) points, read by pigeon_amd.warn_if_untuned
    man = os.path.join(os.environ["MIOPEN_USER_DB_PATH"], "tuned_points.json")
    pts = []
    if os.path.exists(man):
        with open(man) as fh:
            pts = [tuple(p) for p in json.load(fh)]
    pt = (2 * args.batch, args.height // 8, args.width // 8)
    if not args.no_find and pt not in pts:
        pts.append(pt)
        with open(man, "w") as fh:
            json.dump(sorted(pts), fh)
    print(f"done in {time.perf_counter() - t_all:.0f} s; db files: {os.listdir(os.environ['MIOPEN_USER_DB_PATH'])}", flush=True)


if __name__ == "__main__":
    main()
