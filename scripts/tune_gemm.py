#!/usr/bin/env python3
"""Fill a torch TunableOp table for the GEMMs of the hot path (run on the MI355X box).

torch picks hipBLASLt's heuristic solution for every Linear; TunableOp times all hipBLASLt / rocBLAS solutions per
distinct (transA, transB, M, N, K, dtype) once and records the fastest.  The resulting CSV ships in-tree
(pigeon_amd/tunableop/) and pigeon_amd enables TunableOp in look-up mode (no tuning at run time); shapes not in the
table use the heuristic as before.

  python scripts/tune_gemm.py [--batch 16] [--out gpurun_out/tunableop.csv]"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PIGEON_TUNABLEOP"] = "0"   # do not load the shipped table while producing a new one
# Round 2 lost the record of a sweep that ended in a GPU memory fault: its log was overwritten by the next
# run of the same name.  Every run now writes TunableOp's own trace (op signature + candidate before each trial) to a
# file of its own that is flushed line by line, so a run that dies names the last signature / candidate it tried.
os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "1")
_TRACE = os.path.join(ROOT, "gpurun_out", f"tune_gemm_trace_{time.strftime('%m%d_%H%M%S')}_{os.getpid()}.stderr")
os.makedirs(os.path.dirname(_TRACE), exist_ok=True)
_trace_fd = os.open(_TRACE, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
os.dup2(_trace_fd, 2)                  # C++ stderr (TunableOp, hipBLASLt, the HIP runtime's fault message) included
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--out", default="gpurun_out/tunableop.csv")
    ap.add_argument("--max-ms", type=int, default=60)
    ap.add_argument("--unet-only", action="store_true",
                    help="tune the UNet's GEMMs only (the token decoder / VAE keep the heuristic): the sweep over the "
                         "token decoder's small-M shapes faulted inside a library candidate at batch 4")
    args = ap.parse_args()
    t0 = time.perf_counter()

    def hb():
        while True:
            time.sleep(30)
            print(f"[{time.perf_counter() - t0:6.0f}s] tuning...", flush=True)

    threading.Thread(target=hb, daemon=True).start()
    from pigeon_amd import ops, synthetic
    dev = torch.device("cuda", 0)
    pipe = synthetic.build_random_stack(dev, "bf16", seed=123)
    import torch.cuda.tunable as tn
    tn.enable(True)
    tn.tuning_enable(not args.unet_only)
    tn.set_max_tuning_duration(args.max_ms)
    tn.set_max_tuning_iterations(20)
    tn.set_filename(args.out)
    B = args.batch
    h, w = args.height // 8, args.width // 8
    tokens = [t.to(dev) for t in synthetic.synthetic_tokens(B)]
    with torch.no_grad():
        pe, pooled = pipe.generate_image_embeds(tokens)
        print(f"[{time.perf_counter() - t0:6.0f}s] token decoder tuned", flush=True)
        ctx = torch.cat([torch.zeros_like(pe), pe])
        add = torch.cat([torch.zeros_like(pooled), pooled])
        ids = torch.tensor([[args.width, args.height, 0, 0, args.width, args.height]], dtype=torch.bfloat16,
                           device=dev).repeat(2 * B, 1)
        x = ops.cfg_expand_scale(torch.randn(B, 4, h, w, device=dev), torch.bfloat16, 1.0)
        tn.tuning_enable(True)
        pipe.unet(x, 961, encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": add, "time_ids": ids})
        torch.cuda.synchronize()
        # the pipeline runs cross-attention on the conditional half only (HipAttnProcessor.zero_ctx_rows): those
        # projections have half the rows
        pipe.unet.set_zero_context_rows(B)
        pipe.unet(x, 961, encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": add, "time_ids": ids})
        pipe.unet.set_zero_context_rows(0)
        torch.cuda.synchronize()
        print(f"[{time.perf_counter() - t0:6.0f}s] unet tuned", flush=True)
        if args.unet_only:
            tn.tuning_enable(False)
        pipe.decode_latents(torch.randn(1, 4, h, w, device=dev), "uint8")
        torch.cuda.synchronize()
    pass  # the table is written when the process exits
    print(f"[{time.perf_counter() - t0:6.0f}s] {len(tn.get_results())} entries", flush=True)
    # keep the tail of the trace with the judged files (the whole trace stays in gpurun_out/)
    try:
        tag = f"b{args.batch}_{args.height}x{args.width}" + ("_unet_only" if args.unet_only else "")
        with open(_TRACE, errors="replace") as fh:
            tail = fh.readlines()[-300:]
        with open(os.path.join(ROOT, "profiles", f"tune_gemm_{tag}_trace_tail.txt"), "w") as fh:
            fh.write(f"# last {len(tail)} lines of {os.path.basename(_TRACE)} (PYTORCH_TUNABLEOP_VERBOSE=1), sweep completed\n")
            fh.writelines(tail)
    except OSError as e:
        print(f"trace tail not copied: {e}", flush=True)
    tn.tuning_enable(False)
    with torch.no_grad():
        f = lambda: pipe.unet(x, 961, encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": add, "time_ids": ids})
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        print(f"unet fwd with the tuned table: {(time.perf_counter() - t) / 3 * 1e3:.1f} ms", flush=True)
        tn.enable(False)
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        print(f"unet fwd with the library heuristic: {(time.perf_counter() - t) / 3 * 1e3:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
