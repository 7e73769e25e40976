"""Caller side of the hot path: the `generate_images` batching loop and on-disk layout of
Pigeon/inference.py:345-458, plus sample sharding over the GPUs of one node.

Sharding: every (user, target) sample is independent end to end, so ranks take contiguous blocks of the global
sample list; full weight replica per GPU; no collective inside the denoising loop.  The one exchange step is an RCCL
all-gather (torch.distributed backend "nccl" = RCCL over xGMI) of the final latents and per-image scores, once per
outer batch.  Noise: every rank draws the GLOBAL noise tensor from an identically seeded generator and slices its rows
(device generator for the highres decoder, host generator for the lowres one, as the reference draws them), so the
noise rows, token slices and task slices of an N-GPU run are IDENTICAL to the 1-GPU run's; the final latents agree
within the fixed batch-independence bar, not bit for bit (a rank's UNet batch is 1/N of the 1-GPU batch, so the
libraries pick other GEMM / convolution tiles and accumulate in another order: 0.0036 relative L2 per image measured
against a bar of 0.006, tests/test_gpu_runner_dist.py).
"""
import math
import os
import shutil
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of rank `rank`: [r*ceil(M/W), (r+1)*ceil(M/W)) clipped to M."""
    per = math.ceil(total / world) if world > 0 else total
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def global_noise(shape: Sequence[int], seed: int, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The noise the single-GPU reference run would draw for the whole batch (lavit_for_pigeon.py:716-718)."""
    gen = generator if generator is not None else torch.Generator(device=device).manual_seed(seed)
    return torch.randn(tuple(shape), generator=gen, device=device)


def all_gather_results(latents: torch.Tensor, scores: Optional[torch.Tensor], counts: Sequence[int], group=None):
    """All-gather final latents [m_r, 4, h, w] (and scores [m_r, S]) from every rank in ONE collective: each image is
    one fp32 row [4*h*w + S] (latents, then its scores), uneven tails are padded to max(counts) rows for the collective
    and trimmed afterwards.  Works with backend nccl (RCCL over xGMI) and gloo.  Returns (latents, scores | None) in
    global (rank-major = task) order, on the device the inputs live on."""
    if not dist.is_initialized():
        return latents, scores
    world = dist.get_world_size(group)      # a one-rank group still goes through the collective (RCCL rehearsal)
    if latents.shape[0] != counts[dist.get_rank(group)] or (scores is not None and scores.shape[0] != latents.shape[0]):
        raise ValueError("all_gather_results: row counts do not match this rank's entry of `counts`")
    m = max(counts)
    dev = latents.device
    lat_shape, n_lat = tuple(latents.shape[1:]), int(np.prod(latents.shape[1:]))
    n_sc = 0 if scores is None else int(np.prod(scores.shape[1:]))
    host_staged = dist.get_backend(group) == "gloo"   # gloo collectives run on host memory
    cdev = torch.device("cpu") if host_staged else dev
    row = torch.zeros((m, n_lat + n_sc), dtype=torch.float32, device=cdev)
    row[: latents.shape[0], :n_lat] = latents.reshape(latents.shape[0], n_lat).to(cdev, torch.float32)
    if n_sc:
        row[: scores.shape[0], n_lat:] = scores.reshape(scores.shape[0], n_sc).to(cdev, torch.float32)
    out = torch.empty((world * m, n_lat + n_sc), dtype=torch.float32, device=cdev)
    dist.all_gather_into_tensor(out, row, group=group)
    rows = torch.cat([out[r * m: r * m + counts[r]] for r in range(world)], dim=0).to(dev)
    g_lat = rows[:, :n_lat].reshape((rows.shape[0],) + lat_shape).to(latents.dtype)
    g_sc = None if scores is None else rows[:, n_lat:].reshape((rows.shape[0],) + tuple(scores.shape[1:])).to(scores.dtype)
    return g_lat, g_sc


def merge_all_res(dst: Dict, part: Dict) -> Dict:
    """Deep-merge one rank's all_res fragment ({uid: {target: {genre: record}}}, inference.py:421-452) into `dst`."""
    for uid, targets in part.items():
        for tgt, genres in targets.items():
            dst.setdefault(uid, {}).setdefault(tgt, {}).update(genres)
    return dst


def save_batch_results(all_res: Dict, batch_uids, batch_genres, batch_history, batch_target_image_iid, batch_labels,
                       batch_tokens, batch_images, num_return_sequences: int, num_return_images: int,
                       save_folder: str, eval_name: str, img_folder_path: Optional[str] = None,
                       all_image_paths: Optional[List[str]] = None) -> Dict:
    """On-disk layout of inference.py:413-458:
    {save_folder}/{uid}/{target}/{genre}/{seq}/{j}.jpg and all_res[uid][target][genre] = {history, labels, seqs}."""
    for i, uid in enumerate(batch_uids):
        uid = int(uid)
        genre_id = int(batch_genres[i])
        target_iid = int(batch_target_image_iid[i])
        entry = all_res.setdefault(uid, {}).setdefault(target_iid, {})
        rec = entry[genre_id] = {"history": torch.as_tensor(batch_history[i]).clone(),
                                 "labels": torch.as_tensor(batch_labels[i]).clone(), "seqs": {}}
        base = os.path.join(save_folder, str(uid), str(target_iid), str(genre_id))
        os.makedirs(base, exist_ok=True)
        if img_folder_path is not None and all_image_paths is not None:
            for j, iid in enumerate(batch_history[i]):
                iid = int(iid)
                shutil.copy(os.path.join(img_folder_path, all_image_paths[iid]), os.path.join(base, f"{j}-{iid}.jpg"))
        for seq_id in range(num_return_sequences):
            seq = rec["seqs"][seq_id] = {"tokens": batch_tokens[i * num_return_sequences + seq_id].cpu(), "images": []}
            folder = os.path.join(base, str(seq_id))
            os.makedirs(folder, exist_ok=True)
            start = (i * num_return_sequences + seq_id) * num_return_images
            for j, image in enumerate(batch_images[start:start + num_return_images]):
                path = os.path.join(folder, f"{j}.jpg")
                image.save(path)
                seq["images"].append(path.split(f"/{eval_name}/")[-1])
        if img_folder_path is not None and all_image_paths is not None:
            shutil.copy(os.path.join(img_folder_path, all_image_paths[target_iid]), os.path.join(base, "target.jpg"))
    return all_res


def generate_images(lavit, image_tokens: List[torch.Tensor], labels, uids, genres, target_image_iid, history,
                    num_return_sequences: int = 1, batch_size: int = 15, save_folder: Optional[str] = None,
                    eval_name: str = "", img_folder_path: Optional[str] = None,
                    all_image_paths: Optional[List[str]] = None, num_return_images: int = 1, width: int = 512,
                    height: int = 512, original_size=None, crops_coords_top_left=None, num_inference_steps: int = 25,
                    guidance_scale_for_dm: float = 7.0, generator: Optional[torch.Generator] = None,
                    rank: int = 0, world_size: int = 1, seed: int = 123, scorer=None, text_features=None,
                    history_features=None, return_latents: bool = False):
    """inference.py:345-411 with optional rank sharding.  `lavit` is a PigeonPixelDecoder (or any object with the
    reference's generate_image signature).  Each outer batch of `batch_size*num_return_sequences` token sequences is
    split over ranks; every rank writes the JPEGs of its own samples.  After each outer batch the ranks exchange
    (a) ONE all-gather of the final latents and per-image scores (RCCL over xGMI on the GPU, gloo in
    the CPU tests) and (b) their `all_res` fragments, so every rank returns the full `all_res` of the 1-GPU run and
    rank 0 writes `all_res_dm_scale{g}.npy` after every batch exactly as the reference does (inference.py:408).

    With a `scorer` (pigeon_amd.clip_vision.CLIPScorer.score, or any callable(images, text_features [n, D] | None,
    history_features [n, K, D] | None) -> [n, 2]) every rank scores its own images right after generating them
    (cal_scores.py:144-203 fused into the run instead of re-reading the JPEGs); rank 0 writes
    `select_scores_dm_scale{g}.npy` in cal_scores.py's layout.  `text_features` [n_tasks, D] / `history_features`
    [n_tasks, K, D] are per task.

    Latents are collected when `lavit` can hand them over (PigeonPixelDecoder: `output_type="latent"` +
    `decode_latents`); `return_latents=True` requires that and appends the gathered fp32 latents
    [n_images, 4, h/8, w/8] (host memory, global image order, identical on every rank) to the return value.
    Returns all_res, (all_res, select_scores) with a scorer, each followed by the latents when asked for."""
    total_token_num = len(image_tokens)
    step_num = batch_size * num_return_sequences
    total_steps = math.ceil(total_token_num / step_num)
    total_task_num = len(uids)
    all_res: Dict = {}
    gathered_scores: List[torch.Tensor] = []
    gathered_latents: List[torch.Tensor] = []
    per_image = num_return_sequences * num_return_images
    token_start = 0
    device = lavit.device
    lat_c = lavit.unet.config.in_channels
    lat_hw = (height // 8, width // 8)
    has_latents = hasattr(lavit, "decode_latents")
    if return_latents and not has_latents:
        raise ValueError("return_latents needs a pipeline with output_type='latent' and decode_latents")
    # the lowres decoder draws its noise on the HOST (lavit_for_pigeon.py:858-860): keep that stream when sharding
    lowres = getattr(lavit, "pixel_decoding", "highres") == "lowres"
    noise_device = torch.device("cpu") if lowres else device
    if world_size > 1 and generator is None:
        generator = torch.Generator(device=noise_device).manual_seed(seed)   # inference.py:317, identical on every rank
    distributed = world_size > 1
    for step in range(total_steps):
        token_end = min(total_token_num, token_start + step_num)
        start = step * batch_size
        end = min(total_task_num, start + batch_size)
        n_tasks = end - start
        # rank shard, in units of tasks so a task's sequences/images stay on one rank
        lo, hi = shard_range(n_tasks, rank, world_size)
        tok_lo, tok_hi = token_start + lo * num_return_sequences, token_start + hi * num_return_sequences
        latents = None
        if distributed:
            n_img = (token_end - token_start) * num_return_images
            noise = global_noise((n_img, lat_c) + lat_hw, seed, noise_device, generator)
            latents = noise[lo * per_image: hi * per_image]
        batch_res: Dict = {}
        final = torch.empty((0, lat_c) + lat_hw, dtype=torch.float32, device=device)
        if hi > lo:
            batch_tokens = [image_tokens[i].to(device) for i in range(tok_lo, tok_hi)]
            kwargs = dict(image_tokens=batch_tokens, num_return_images=num_return_images, width=width, height=height,
                          original_size=original_size, crops_coords_top_left=crops_coords_top_left,
                          num_inference_steps=num_inference_steps, guidance_scale_for_diffusion=guidance_scale_for_dm,
                          generator=generator)
            if latents is not None:
                kwargs["latents"] = latents
            if has_latents:
                final = lavit.generate_image(output_type="latent", **kwargs)[0].to(torch.float32)
                batch_images = lavit.decode_latents(final, "pil")
            else:
                batch_images, _, _ = lavit.generate_image(**kwargs)
            if save_folder is not None:
                image_save_folder = os.path.join(save_folder, f"dm_scale{guidance_scale_for_dm}")
                os.makedirs(image_save_folder, exist_ok=True)
                save_batch_results(batch_res, uids[start + lo:start + hi], genres[start + lo:start + hi],
                                   history[start + lo:start + hi], target_image_iid[start + lo:start + hi],
                                   labels[start + lo:start + hi], batch_tokens, batch_images, num_return_sequences,
                                   num_return_images, image_save_folder, eval_name, img_folder_path, all_image_paths)
        sc = None
        if scorer is not None:
            if hi > lo:
                tf = None if text_features is None else \
                    torch.as_tensor(text_features[start + lo:start + hi]).repeat_interleave(per_image, dim=0)
                hf = None if history_features is None else \
                    torch.as_tensor(history_features[start + lo:start + hi]).repeat_interleave(per_image, dim=0)
                sc = torch.as_tensor(scorer(batch_images, tf, hf), dtype=torch.float32).to(device)
            else:
                sc = torch.empty(0, 2, dtype=torch.float32, device=device)
        if distributed:
            counts = []
            for r in range(world_size):
                a, b = shard_range(n_tasks, r, world_size)
                counts.append((b - a) * per_image)
            if has_latents or sc is not None:
                # without latents (a foreign pipeline) a one-column dummy keeps the row layout
                carry = final if has_latents else torch.zeros((counts[rank], 1), dtype=torch.float32, device=device)
                final, sc = all_gather_results(carry, sc, counts)
            if save_folder is not None:
                parts = [None] * world_size
                dist.all_gather_object(parts, batch_res)
                for part in parts:              # rank order = task order (contiguous shards)
                    merge_all_res(all_res, part)
        else:
            merge_all_res(all_res, batch_res)
        if save_folder is not None and rank == 0:
            np.save(os.path.join(save_folder, f"all_res_dm_scale{guidance_scale_for_dm}.npy"), np.array(all_res))
        if sc is not None:
            gathered_scores.append(sc.cpu())
        if return_latents:
            gathered_latents.append(final.cpu())
        token_start += batch_size * num_return_sequences
    out = [all_res]
    if scorer is not None:
        scores = torch.cat(gathered_scores, dim=0) if gathered_scores else torch.empty(0, 2)
        select = select_scores_from_lists(uids, target_image_iid, genres, scores, num_return_sequences, num_return_images)
        if save_folder is not None and rank == 0:
            np.save(os.path.join(save_folder, f"select_scores_dm_scale{guidance_scale_for_dm}.npy"), np.array(select))
        out.append(select)
    if return_latents:
        out.append(torch.cat(gathered_latents, dim=0) if gathered_latents
                   else torch.empty((0, lat_c) + lat_hw, dtype=torch.float32))
    return out[0] if len(out) == 1 else tuple(out)


def select_scores_from_lists(uids, target_image_iid, genres, scores: torch.Tensor, num_return_sequences: int,
                             num_return_images: int) -> Dict:
    """cal_scores.py:184-204: select_scores[uid][target][genre][seq] = {"tgt_score": [per image j], "hist_score": [...]}
    from per-image rows ordered task-major, then sequence, then image (inference.py:447-448)."""
    if scores.shape[0] != len(uids) * num_return_sequences * num_return_images:
        raise ValueError("scores do not cover the task list")
    out: Dict = {}
    for i, uid in enumerate(uids):
        slot = out.setdefault(int(uid), {}).setdefault(int(target_image_iid[i]), {}).setdefault(int(genres[i]), {})
        for seq in range(num_return_sequences):
            rows = scores[(i * num_return_sequences + seq) * num_return_images:
                          (i * num_return_sequences + seq + 1) * num_return_images]
            slot[seq] = {"tgt_score": [float(v) for v in rows[:, 0]], "hist_score": [float(v) for v in rows[:, 1]]}
    return out
