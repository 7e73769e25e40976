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
  kernels       the same per kernel class (attention TFLOP/s; GroupNorm / LayerNorm / step ... GB/s with their fraction
                of the 8 TB/s HBM peak)
  cpu_baseline  the oracle (fp32 torch CPU restatement, kind "port") timed on this box's host cores on a bounded sample
                (token stage + 2 denoising steps, each reported; `cores` = threads torch actually ran)
--config i selects the per-GPU share of BASELINE.json configs[i] (1: default; 2: 8 x 1216x832 x 50 steps; 3: 2 token
sequences x 8 candidates; 4: 2048x2048), so an N-GPU driver run can use BASELINE's own multi-GPU workloads.
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

# Per-GPU shares of BASELINE.json `configs`; configs[0] is the CPU case (the cpu_baseline leg)
CONFIG_PRESETS = {
    1: dict(batch=16, height=1024, width=1024, denoise_steps=25, num_return_images=1,
            name="configs[1]: SER-30K-style sticker batch 16 per GPU, 1024x1024, 25 DDIM steps"),
    2: dict(batch=8, height=1216, width=832, denoise_steps=50, num_return_images=1,
            name="configs[2]: MovieLens-style poster batch 64 over 8 GPUs = 8 per GPU, 1216x832 (ratio 2:3), 50 DDIM steps"),
    3: dict(batch=16, height=1024, width=1024, denoise_steps=25, num_return_images=8,
            name="configs[3]: multi-candidate generation, 2 token sequences x 8 candidates = 16 images per GPU, "
                 "1024x1024, 25 DDIM steps"),
    4: dict(batch=2, height=2048, width=2048, denoise_steps=25, num_return_images=1,
            name="configs[4]: high-resolution decode, 2x latent side (2048x2048 images, self-attention N = 16384), "
                 "batch 2 per GPU, 25 DDIM steps"),
}

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


def host_threads(requested):
    """Threads the CPU baseline may use: the request, else OMP_NUM_THREADS (the box's CPU share), else the affinity
    mask, never more than the CPUs this process may run on."""
    allowed = len(os.sched_getaffinity(0))
    want = requested or int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or allowed
    return max(1, min(want, allowed)), allowed


def cpu_baseline(pipe, tokens, height, width, n_steps, guidance, threads, timed_steps=2):
    """Oracle (fp32, CPU) on a bounded sample of configs[0]: the token stage + `timed_steps` of the n_steps denoising
    steps (CFG pair) for one image, each timed, extrapolated to n_steps with their mean.
    Returns (images/sec, token-stage seconds, [seconds per timed step], threads actually used)."""
    from oracle import models_ref, ops_ref, pipeline_ref
    from oracle.models_ref import Policy
    torch.set_num_threads(threads)
    threads = torch.get_num_threads()
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
        t_steps = []
        for t in sched.timesteps[:timed_steps]:
            t0 = time.perf_counter()
            eps = models_ref.unet_forward(pol, unet_sd, cfg, torch.cat([lat] * 2), t, ctx,
                                          {"text_embeds": add, "time_ids": ids})
            lat = sched.step(ops_ref.cfg_combine_ref(eps, guidance), t, lat)[0]
            t_steps.append(time.perf_counter() - t0)
    total = t_embed + n_steps * sum(t_steps) / len(t_steps)
    return 1.0 / total, t_embed, t_steps, threads


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, script=None, poll_s=0.2, env=None):
    """`python bench.py --gpus N` from a plain shell (no WORLD_SIZE): start N fresh rank processes of this script, one
    per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, as torch.distributed.run
    sets them), relay rank 0's stdout (the one JSON line), send the other ranks' stdout to stderr, and return the first
    non-zero exit code of any rank (its siblings are then stopped by PID: a rank that died would leave them waiting in
    the rendezvous).  The parent never touches the GPU: nothing is re-executed after a HIP call.
    The reference's own multi-GPU switch is the broken WORLD_SIZE branch of Pigeon/inference.py:120-124."""
    import subprocess
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, "-u", script or os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), text=r == 0))

    def relay():   # line by line; only rank 0's JSON line goes to stdout (gloo / RCCL banners land there too)
        for ln in procs[0].stdout:
            out = sys.stdout if ln.lstrip().startswith("{") else sys.stderr
            out.write(ln)
            out.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        time.sleep(poll_s)
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code if code > 0 else 128 - code
                    print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
    for r in live:                      # only after a failure: exact PIDs of our own children
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    t.join(timeout=10)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIG_PRESETS),
                    help="BASELINE.json configs[i] preset (per-GPU share); explicit flags below override it")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step")
    ap.add_argument("--denoise-steps", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--guidance", type=float, default=7.0)
    ap.add_argument("--num-return-images", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU baseline (0: OMP_NUM_THREADS, else the affinity mask)")
    ap.add_argument("--cpu-steps", type=int, default=2, help="denoising steps the CPU baseline times (each reported)")
    ap.add_argument("--cross-kv-cache", action="store_true", help="reuse cross-attention K/V across denoising steps")
    ap.add_argument("--hip-graph", action="store_true",
                    help="replay the denoising loop as one hipGraph (pays when the loop is launch-bound: batch <= 2)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real path) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal of the multi-rank control flow on a one-GPU box: every rank uses cuda:0 (gloo only)")
    args = ap.parse_args()
    preset = CONFIG_PRESETS[args.config]
    for k in ("batch", "denoise_steps", "height", "width", "num_return_images"):
        if getattr(args, k) is None:
            setattr(args, k, preset[k])

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (before any GPU call or torch.cuda query)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        world = dist.get_world_size()       # n_gpus of the line = ranks that really joined

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
    # The global-noise contract of runner.generate_images: the N ranks work on ONE job of N x batch
    # images - one token list, one generator seeded alike on every rank that draws the job's whole noise tensor per
    # step; a rank takes its contiguous rows of both.  The images are those of the same job on one GPU (rank 0's rows:
    # bit for bit those of the N = 1 bench run, same batch shape, same kernels).
    tokens = [t.to(device) for t in synthetic.synthetic_tokens(n_seq * world, seed=123)[rank * n_seq:(rank + 1) * n_seq]]
    gen = torch.Generator(device=device).manual_seed(123)
    lat_shape = (args.batch * world, pipe.unet.config.in_channels, args.height // 8, args.width // 8)
    timer = LaunchTimer()
    ops.set_launch_timer(timer)

    loop_marks = []   # (start, end-of-denoising-loop) event pairs of the timed steps: SURVEY 8(d) metric (i)

    def one_step():
        mark = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
        last["mark"] = mark
        mark[0].record()
        noise = runner.global_noise(lat_shape, 123, device, gen)[rank * args.batch:(rank + 1) * args.batch]
        images, _, _ = pipe.generate_image(image_tokens=tokens, num_return_images=args.num_return_images,
                                           width=args.width, height=args.height,
                                           num_inference_steps=args.denoise_steps,
                                           guidance_scale_for_diffusion=args.guidance, latents=noise,
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
            # HBM-bound kernels: algorithmic bytes / summed event time against the 8 TB/s HBM3E peak
            kernels[kind] = {"launches": n, "avg_us": secs / n * 1e6, "GB/s": work / secs / 1e9,
                             "bound": "hbm", "peak_GB/s": PEAK_HBM_GBS, "frac": work / secs / 1e9 / PEAK_HBM_GBS,
                             "share_of_step": secs / (elapsed / args.steps)}
    roofline = None
    att = {k: agg[k] for k in ("attention_w64", "attention") if k in agg}
    if att:
        # dominant hand-written kernel: the 64-rows-per-wave kernel (at batch 16 every head-dim-64 launch of the UNet,
        # self- and cross-attention); `all_head_dim_64` adds the launches the heuristic leaves to the 32-row kernel
        # (small batches), launch-weighted
        dom = "attention_w64" if "attention_w64" in att else "attention"
        n, work, secs = att[dom]
        ach = work / secs / 1e12
        # (full key tiles, as at every SDXL shape: the ring form of the 64-row kernel)
        name = {"attention_w64": "pgn_attn_w64r_kernel<1, true, false, 2, false>", "attention": "pgn_attn_fwd_kernel<1, false, 4, 3, 64, false>"}[dom]
        traffic, src = pmc_traffic(name)
        if traffic is None and dom == "attention_w64":   # summaries taken before the kernel got its later template parameters
            traffic, src = pmc_traffic("pgn_attn_w64r_kernel<1")
        n_all = sum(v[0] for v in att.values())
        w_all = sum(v[1] for v in att.values())
        s_all = sum(v[2] for v in att.values())
        roofline = {"kernel": name + " (bf16, head dim 64)", "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": src, "launches": n, "avg_launch_us": secs / n * 1e6,
                    "flop_per_launch": work / n,
                    "all_head_dim_64": {"launches": n_all, "avg_launch_us": s_all / n_all * 1e6,
                                        "achieved": w_all / s_all / 1e12, "frac": w_all / s_all / 1e12 / PEAK_BF16_TFLOPS,
                                        "flop_per_launch": w_all / n_all}}

    state["phase"] = "cpu baseline"
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads, allowed = host_threads(args.cpu_threads)
        ips, t_embed, t_steps, used = cpu_baseline(pipe, [t.cpu() for t in tokens], args.height, args.width,
                                                   args.denoise_steps, args.guidance, threads, args.cpu_steps)
        cpu = {"value": ips, "unit": "images/sec", "cores": used, "kind": "port",
               "host_cpus_allowed": allowed, "token_stage_s": t_embed, "step_s": t_steps,
               "extrapolated_total_s": 1.0 / ips,
               "sample": f"configs[0]: 1 image {args.width}x{args.height} fp32 on {used} host threads, token stage "
                         f"({t_embed:.1f} s) + {len(t_steps)} of {args.denoise_steps} CFG denoising steps timed "
                         f"({', '.join(f'{x:.1f}' for x in t_steps)} s), extrapolated to {args.denoise_steps} steps "
                         "with their mean; VAE excluded"}

    if rank == 0:
        line = {
            "metric": "images/sec at 25 steps w/ CFG, 1/2/4/8 GPUs; attention MFMA util %",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{preset['name']} [as run: batch {args.batch} per GPU, {args.width}x{args.height}, "
                                   f"{args.denoise_steps} DDIM steps, {args.num_return_images} image(s) per token "
                                   f"sequence] with CFG {args.guidance}, bf16, synthetic tokens, "
                                   "random-init SDXL-shaped decoder (UNet 2.57B + token decoder 414M + VAE); "
                                   "end to end incl. VAE decode and uint8 copy to host",
                       "baseline_config": args.config, "batch_per_gpu": args.batch,
                       "num_return_images": args.num_return_images, "height": args.height, "width": args.width,
                       "denoise_steps": args.denoise_steps, "parallelism": f"sample-sharded dp{world}",
                       "noise": "one generator (seed 123) draws the job's global noise on every rank, rows sliced by rank",
                       "dist_backend": args.dist_backend if world > 1 else None,
                       # the version-keyed tuning tables were accepted by the running libraries (else the hipBLASLt /
                       # MIOpen heuristics ran and the constructor has warned)
                       "tuned_gemms": bool(pipe.tuned_gemms), "miopen_db_hit": bool(pipe.miopen_db_hit)},
            # ranks that joined the RCCL (backend "nccl") process group; 0 for a one-rank run or a gloo rehearsal
            "rccl_ranks": world if (world > 1 and args.dist_backend == "nccl") else 0,
            # peak of torch's allocator on this rank over build + warm-up + timed steps (weights, activations, library
            # workspaces it serves); of 288 GB per MI355X
            "hbm_peak_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
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
