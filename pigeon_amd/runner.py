"""Caller side of the hot path: the `generate_images` batching loop and on-disk layout of
Pigeon/inference.py:345-458, plus sample sharding over the GPUs of one node.

Sharding: every (user, target) sample is independent end to end, so ranks take contiguous blocks of the global
sample list; full weight replica per GPU; no collective inside the denoising loop.  The one exchange step is an RCCL
all-gather (torch.distributed backend "nccl" = RCCL over xGMI) of the final latents and per-image scores, once per
outer batch.  Noise: every rank draws the GLOBAL noise tensor from an identically seeded generator and slices its rows,
so an N-GPU run reproduces the 1-GPU run bit for bit.
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
    """All-gather final latents [m_r, 4, h, w] (and scores [m_r, S]) from every rank; uneven tails are padded to
    max(counts) rows for the collective and trimmed afterwards.  Works with backend nccl (RCCL) and gloo."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return latents, scores
    m = max(counts)

    host_staged = dist.get_backend(group) == "gloo"   # gloo collectives run on host memory

    def gather(t):
        dev = t.device
        if host_staged:
            t = t.cpu()
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        out = torch.empty((world * m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
        return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)], dim=0).to(dev)

    return gather(latents), (None if scores is None else gather(scores))


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
                    history_features=None):
    """inference.py:345-411 with optional rank sharding.  `lavit` is a PigeonPixelDecoder (or any object with the
    reference's generate_image signature).  Each outer batch of `batch_size*num_return_sequences` token sequences is
    split over ranks; every rank writes the JPEGs of its own samples and returns its part of `all_res`
    (rank 0 of a single-GPU run writes `all_res_dm_scale{g}.npy` exactly as the reference does).

    With a `scorer` (pigeon_amd.clip_vision.CLIPScorer.score, or any callable(images, text_features [n, D] | None,
    history_features [n, K, D] | None) -> [n, 2]) every rank scores its own images right after generating them
    (cal_scores.py:144-203 fused into the run instead of re-reading the JPEGs); the per-image (tgt_score, hist_score)
    rows are all-gathered once per outer batch and rank 0 writes `select_scores_dm_scale{g}.npy`
    in cal_scores.py's layout.  `text_features` [n_tasks, D] / `history_features` [n_tasks, K, D] are per task.
    Returns all_res, or (all_res, select_scores) when a scorer is given."""
    total_token_num = len(image_tokens)
    step_num = batch_size * num_return_sequences
    total_steps = math.ceil(total_token_num / step_num)
    total_task_num = len(uids)
    all_res: Dict = {}
    gathered_scores: List[torch.Tensor] = []
    per_image = num_return_sequences * num_return_images
    token_start = 0
    device = lavit.device
    lat_c = lavit.unet.config.in_channels
    if world_size > 1 and generator is None:
        generator = torch.Generator(device=device).manual_seed(seed)   # inference.py:317, identical on every rank
    for step in range(total_steps):
        token_end = min(total_token_num, token_start + step_num)
        start = step * batch_size
        end = min(total_task_num, start + batch_size)
        n_tasks = end - start
        # rank shard, in units of tasks so a task's sequences/images stay on one rank
        lo, hi = shard_range(n_tasks, rank, world_size)
        tok_lo, tok_hi = token_start + lo * num_return_sequences, token_start + hi * num_return_sequences
        latents = None
        if world_size > 1:
            n_img = (token_end - token_start) * num_return_images
            noise = global_noise((n_img, lat_c, height // 8, width // 8), seed, device, generator)
            latents = noise[lo * num_return_sequences * num_return_images: hi * num_return_sequences * num_return_images]
        if hi > lo:
            batch_tokens = [image_tokens[i].to(device) for i in range(tok_lo, tok_hi)]
            kwargs = dict(image_tokens=batch_tokens, num_return_images=num_return_images, width=width, height=height,
                          original_size=original_size, crops_coords_top_left=crops_coords_top_left,
                          num_inference_steps=num_inference_steps, guidance_scale_for_diffusion=guidance_scale_for_dm,
                          generator=generator)
            if latents is not None:
                kwargs["latents"] = latents
            batch_images, _, _ = lavit.generate_image(**kwargs)
            if save_folder is not None:
                image_save_folder = os.path.join(save_folder, f"dm_scale{guidance_scale_for_dm}")
                os.makedirs(image_save_folder, exist_ok=True)
                all_res = save_batch_results(all_res, uids[start + lo:start + hi], genres[start + lo:start + hi],
                                             history[start + lo:start + hi], target_image_iid[start + lo:start + hi],
                                             labels[start + lo:start + hi], batch_tokens, batch_images,
                                             num_return_sequences, num_return_images, image_save_folder, eval_name,
                                             img_folder_path, all_image_paths)
                suffix = "" if world_size == 1 else f".rank{rank}"
                np.save(os.path.join(save_folder, f"all_res_dm_scale{guidance_scale_for_dm}{suffix}.npy"),
                        np.array(all_res))
        if scorer is not None:
            if hi > lo:
                tf = None if text_features is None else \
                    torch.as_tensor(text_features[start + lo:start + hi]).repeat_interleave(per_image, dim=0)
                hf = None if history_features is None else \
                    torch.as_tensor(history_features[start + lo:start + hi]).repeat_interleave(per_image, dim=0)
                sc = torch.as_tensor(scorer(batch_images, tf, hf), dtype=torch.float32)
            else:
                sc = torch.empty(0, 2, dtype=torch.float32)
            if world_size > 1:
                counts = []
                for r in range(world_size):
                    a, b = shard_range(n_tasks, r, world_size)
                    counts.append((b - a) * per_image)
                dev_sc = sc.to(device) if dist.get_backend() != "gloo" else sc.cpu()
                _, sc = all_gather_results(dev_sc.new_zeros((dev_sc.shape[0], 1)), dev_sc, counts)
            gathered_scores.append(sc.cpu())
        token_start += batch_size * num_return_sequences
    if scorer is None:
        return all_res
    scores = torch.cat(gathered_scores, dim=0) if gathered_scores else torch.empty(0, 2)
    select = select_scores_from_lists(uids, target_image_iid, genres, scores, num_return_sequences, num_return_images)
    if save_folder is not None and rank == 0:
        np.save(os.path.join(save_folder, f"select_scores_dm_scale{guidance_scale_for_dm}.npy"), np.array(select))
    return all_res, select


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
