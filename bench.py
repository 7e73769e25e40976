#!/usr/bin/env python3
"""bench.py — images/sec of the Pigeon image-generation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over one batch: generate_image (token gather + token decoder, 25 DDIM steps
with CFG through the SDXL-shaped UNet, VAE decode, uint8 images copied to the host) for BASELINE.json configs[1]:
batch 16 at 1024x1024 per GPU, bf16, synthetic tokens, random-init weights (no network: no checkpoint).  Samples shard
over ranks with no collective in the loop; each rank all-gathers its final latents once per step (weak scaling).

The JSON line carries, besides the driver's contract fields:
  roofline      the dominant hand-written kernel (MFMA attention): algorithmic FLOPs of every attention launch in the
                timed region / their summed HIP-event durations (events recorded on the launch stream)
  kernels       the same per kernel class (attention TFLOP/s; GroupNorm / step GB/s)
  cpu_baseline  the oracle (fp32 torch CPU restatement, kind "port") timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import threading

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0       # HBM3E peak


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*_pmc.json, written by
    scripts/profile_bench.sh + scripts/profile_summary.py: separate FETCH_SIZE / WRITE_SIZE passes over this same
    command, gfx950 FETCH_SIZE x2 correction).  Counters cannot be collected from inside the timed run, so the figure
    is read from the file and its name is reported beside it; None when no summary exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        data = json.load(fh)
    for name, e in data.items():
        if kernel in name and "hbm_bytes_per_launch" in e:
            return e["hbm_bytes_per_launch"]["total"], os.path.relpath(files[-1], ROOT)
    return None, None


class LaunchTimer:
    """HIP-event bracket around every C-ABI launch (events on torch's current stream = the launch stream)."""

    def __init__(self):
        self.records = []
        self.enabled = False

    def __call__(self, kind, work, fn):
        if not self.enabled:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.records.append((kind, work, s, e))
        return out

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for kind, work, s, e in self.records:
            a = agg.setdefault(kind, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += work
            a[2] += s.elapsed_time(e) * 1e-3
        return agg


def cpu_baseline(pipe, tokens, height, width, n_steps, guidance, threads):
    """Oracle (fp32, CPU) on a bounded sample of configs[0]: the token stage + ONE of the n_steps denoising steps
    (CFG pair) for one image, extrapolated to n_steps.  Returns images/sec."""
    from oracle import models_ref, ops_ref, pipeline_ref
    from oracle.models_ref import Policy
    torch.set_num_threads(threads)
    pol = Policy(torch.float32)
    f32 = lambda m: {k: v.detach().to("cpu", torch.float32) for k, v in m.state_dict().items()}
    unet_sd, dec_sd = f32(pipe.unet), f32(pipe.tokenizer_decoder)
    cfg = dict(pipe.unet.config)
    codebook = pipe.codebook.cpu()
    with torch.no_grad():
        t0 = time.perf_counter()
        emb, nums = ops_ref.token_gather_ref(tokens[:1], codebook, torch.float32)
        pe, pooled = models_ref.token_decoder_forward(pol, dec_sd, emb, nums)
        t_embed = time.perf_counter() - t0
        sched = pipeline_ref.DDIMRef(**{k: v for k, v in pipe.scheduler.config.items()})
        sched.set_timesteps(n_steps)
        lat = torch.randn(1, 4, height // 8, width // 8, generator=torch.Generator().manual_seed(123))
        ids = torch.tensor([[width, height, 0, 0, width, height]], dtype=torch.float32).repeat(2, 1)
        ctx = torch.cat([torch.zeros_like(pe), pe])
        add = torch.cat([torch.zeros_like(pooled), pooled])
        t0 = time.perf_counter()
        t = sched.timesteps[0]
        eps = models_ref.unet_forward(pol, unet_sd, cfg, torch.cat([lat] * 2), t, ctx, {"text_embeds": add, "time_ids": ids})
        lat = sched.step(ops_ref.cfg_combine_ref(eps, guidance), t, lat)[0]
        t_step = time.perf_counter() - t0
    total = t_embed + n_steps * t_step
    return 1.0 / total, t_embed, t_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step (configs[1]: 16)")
    ap.add_argument("--denoise-steps", type=int, default=25)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--guidance", type=float, default=7.0)
    ap.add_argument("--num-return-images", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--cross-kv-cache", action="store_true", help="reuse cross-attention K/V across denoising steps")
    ap.add_argument("--hip-graph", action="store_true",
                    help="replay the denoising loop as one hipGraph (pays when the loop is launch-bound: batch <= 2)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real path) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal of the multi-rank control flow on a one-GPU box: every rank uses cuda:0 (gloo only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.single_device:
        if args.dist_backend == "nccl" and world > 1:
            raise SystemExit("--single-device shares one GPU between ranks: RCCL refuses that, use --dist-backend gloo")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.dist_backend)

    t_start = time.perf_counter()
    state = {"phase": "build"}

    def heartbeat():   # keeps log watchers alive during long silent phases; stderr only
        while state["phase"] != "done":
            time.sleep(30)
            if rank == 0 and state["phase"] != "done":
                print(f"[bench {time.perf_counter() - t_start:6.0f}s] {state['phase']}", file=sys.stderr, flush=True)

    threading.Thread(target=heartbeat, daemon=True).start()

    from pigeon_amd import ops, runner, synthetic
    pipe = synthetic.build_random_stack(device, "bf16", seed=123)
    if args.cross_kv_cache:
        pipe.unet.enable_cross_kv_cache(True)
    if args.hip_graph:
        if world > 1:
            raise SystemExit("--hip-graph is the single-GPU small-batch latency option (no per-step callback, so no "
                             "latents for the all-gather)")
        pipe.enable_hip_graph(True)
    n_seq = args.batch // args.num_return_images
    tokens = [t.to(device) for t in synthetic.synthetic_tokens(n_seq, seed=123 + rank)]
    gen = torch.Generator(device=device).manual_seed(123 + rank)
    timer = LaunchTimer()
    ops.set_launch_timer(timer)

    loop_marks = []   # (start, end-of-denoising-loop) event pairs of the timed steps: SURVEY 8(d) metric (i)

    def one_step():
        mark = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
        last["mark"] = mark
        mark[0].record()
        images, _, _ = pipe.generate_image(image_tokens=tokens, num_return_images=args.num_return_images,
                                           width=args.width, height=args.height,
                                           num_inference_steps=args.denoise_steps,
                                           guidance_scale_for_diffusion=args.guidance, generator=gen,
                                           output_type="uint8", callback=None if args.hip_graph else keep_latents)
        host = images.cpu()   # device -> host boundary (lavit_for_pigeon.py:782), uint8 instead of fp32
        if world > 1:         # the path's one exchange step: collect final latents from every rank
            runner.all_gather_results(last["latents"], None, [args.batch] * world)
        return host

    last = {}

    def keep_latents(i, t, x):
        last["latents"] = x
        if i == args.denoise_steps - 1:          # the loop has been enqueued: what follows is VAE decode + uint8 copy
            last["mark"][1].record()
            loop_marks.append(last["mark"])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    state["phase"] = "warmup"
    for _ in range(args.warmup):
        one_step()
    fence()
    loop_marks.clear()
    state["phase"] = "timed steps"
    # HIP-event brackets around every C-ABI launch cost ~1 % of a step (25 k event pairs), so they are recorded during
    # the LAST timed step only: still inside the timed region, on the launch stream, at 1/K of the cost
    t0 = time.perf_counter()
    for i in range(args.steps):
        timer.enabled = (not args.no_kernel_timing) and i == args.steps - 1
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    tt = torch.tensor([elapsed], device=device if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())

    images_total = args.batch * world * args.steps
    value = images_total / elapsed
    loop_ms = sum(a.elapsed_time(b) for a, b in loop_marks) / max(1, len(loop_marks))   # this rank, per step
    agg = timer.summary() if not args.no_kernel_timing else {}
    kernels = {}
    for kind, (n, work, secs) in agg.items():
        if secs <= 0:
            continue
        if kind.startswith("attention"):
            kernels[kind] = {"launches": n, "avg_us": secs / n * 1e6, "TFLOP/s": work / secs / 1e12,
                             "share_of_step": secs / (elapsed / args.steps)}
        else:
            kernels[kind] = {"launches": n, "avg_us": secs / n * 1e6, "GB/s": work / secs / 1e9,
                             "share_of_step": secs / (elapsed / args.steps)}
    roofline = None
    if "attention" in agg:
        n, work, secs = agg["attention"]
        ach = work / secs / 1e12
        # the UNet's instantiation (bf16, no key bias, 4 waves, head dim 64); other head dims are timed as their own kind
        traffic, src = pmc_traffic("pgn_attn_fwd_kernel<1, false, 4, 3, 64, false>")
        roofline = {"kernel": "pgn_attn_fwd_kernel<bf16, head_dim 64>", "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": src, "launches": n, "avg_launch_us": secs / n * 1e6,
                    "flop_per_launch": work / n}

    state["phase"] = "cpu baseline"
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ips, t_embed, t_step = cpu_baseline(pipe, [t.cpu() for t in tokens], args.height, args.width,
                                            args.denoise_steps, args.guidance, args.cpu_threads)
        cpu = {"value": ips, "unit": "images/sec", "cores": args.cpu_threads, "kind": "port",
               "sample": f"configs[0]: 1 image {args.width}x{args.height} fp32, token stage ({t_embed:.1f} s) + 1 of "
                         f"{args.denoise_steps} CFG denoising steps ({t_step:.1f} s) timed, extrapolated x{args.denoise_steps}; "
                         "VAE excluded"}

    if rank == 0:
        line = {
            "metric": "images/sec at 25 steps w/ CFG, 1/2/4/8 GPUs; attention MFMA util %",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"configs[1]: SER-30K-style sticker batch {args.batch} per GPU, {args.width}x{args.height}, "
                                   f"{args.denoise_steps} DDIM steps with CFG {args.guidance}, bf16, synthetic tokens, "
                                   "random-init SDXL-shaped decoder (UNet 2.57B + token decoder 414M + VAE); "
                                   "end to end incl. VAE decode and uint8 copy to host",
                       "batch_per_gpu": args.batch, "height": args.height, "width": args.width,
                       "denoise_steps": args.denoise_steps, "parallelism": f"sample-sharded dp{world}"},
            # embeds + denoising loop only (no VAE decode / uint8 copy), this rank's GPU time per step
            "loop_only": {"images_per_sec_per_gpu": args.batch / (loop_ms * 1e-3) if loop_ms > 0 else None,
                          "ms_per_step": loop_ms if loop_ms > 0 else None},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
            "attention_mfma_util_pct": None if roofline is None else 100.0 * roofline["frac"],
        }
        print(json.dumps(line), flush=True)
    state["phase"] = "done"
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
