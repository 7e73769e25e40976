"""Scheduler + generate_image oracle (CPU).  TEST INFRASTRUCTURE ONLY.

`DDIMRef` / `EulerRef` restate diffusers 0.27.2 DDIMScheduler / EulerDiscreteScheduler (PARITY UNPINNED: third-party,
absent from /root/reference; call sites lavit_for_pigeon.py:189,719-720,748,764).
`generate_image_ref` restates the control flow of Pigeon/models/lavit_for_pigeon.py:706-764 (+ :766-782 when a VAE
state dict is given) line by line over the functional models of oracle.models_ref.
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import models_ref, ops_ref
from .models_ref import Policy


def make_betas(schedule, beta_start, beta_end, T):
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
    raise NotImplementedError(schedule)


class DDIMRef:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 clip_sample_range=1.0, timestep_spacing="leading", **_):
        self.T, self.clip_sample, self.clip_range = num_train_timesteps, clip_sample, clip_sample_range
        self.offset, self.pred, self.spacing = steps_offset, prediction_type, timestep_spacing
        self.alphas_cumprod = torch.cumprod(1.0 - make_betas(beta_schedule, beta_start, beta_end, self.T), dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.n = None

    def set_timesteps(self, n):
        self.n = n
        if self.spacing == "linspace":
            ts = np.linspace(0, self.T - 1, n).round()[::-1].copy().astype(np.int64)
        elif self.spacing == "leading":
            ts = (np.arange(0, n) * (self.T // n)).round()[::-1].copy().astype(np.int64) + self.offset
        elif self.spacing == "trailing":
            ts = np.round(np.arange(self.T, 0, -self.T / n)).astype(np.int64) - 1
        else:
            raise ValueError(self.spacing)
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, t=None):
        return sample

    def step(self, model_output, t, sample, use_clipped_model_output=False):
        t = int(t)
        prev_t = t - self.T // self.n
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        clip = self.clip_range if self.clip_sample else None
        return ops_ref.ddim_step_ref(model_output, sample, a_t, a_prev, self.pred, clip, use_clipped_model_output)


def threshold_sample_ref(sample: torch.Tensor, ratio: float, sample_max_value: float) -> torch.Tensor:
    """diffusers 0.27.2 DDIMScheduler._threshold_sample ("dynamic thresholding", Imagen): per sample, s = the `ratio`
    quantile of |x0| (linear interpolation between order statistics), clamped to [1, sample_max_value]; x0 is clipped
    to [-s, s] and divided by s.  Restated with an explicit sort so it does not share torch.quantile with the product.
    PARITY UNPINNED (third-party; not on the reference's default path: thresholding is off in SDXL's scheduler)."""
    b = sample.shape[0]
    flat = sample.double().reshape(b, -1)
    a = np.sort(np.abs(flat.numpy()), axis=1)
    n = a.shape[1]
    pos = ratio * (n - 1)
    lo, hi = int(math.floor(pos)), min(int(math.floor(pos)) + 1, n - 1)
    s = a[:, lo] + (a[:, hi] - a[:, lo]) * (pos - lo)
    s = np.clip(s, 1.0, sample_max_value)[:, None]
    out = np.clip(flat.numpy(), -s, s) / s
    return torch.from_numpy(out).reshape(sample.shape).to(sample.dtype)


class EulerRef:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 prediction_type="epsilon", timestep_spacing="linspace", steps_offset=0, **_):
        self.T, self.pred, self.spacing, self.offset = num_train_timesteps, prediction_type, timestep_spacing, steps_offset
        self.alphas_cumprod = torch.cumprod(1.0 - make_betas(beta_schedule, beta_start, beta_end, self.T), dim=0)
        s = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        self.sigmas = torch.cat([torch.from_numpy(s[::-1].copy()).float(), torch.zeros(1)])
        self.idx = None

    @property
    def init_noise_sigma(self):
        m = self.sigmas.max()
        return m if self.spacing in ("linspace", "trailing") else (m ** 2 + 1) ** 0.5

    def set_timesteps(self, n):
        T = self.T
        if self.spacing == "linspace":
            ts = np.linspace(0, T - 1, n, dtype=np.float32)[::-1].copy()
        elif self.spacing == "leading":
            ts = (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.float32) + self.offset
        else:
            ts = (np.arange(T, 0, -T / n)).round().copy().astype(np.float32) - 1
        s = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        s = np.interp(ts, np.arange(0, len(s)), s)
        self.sigmas = torch.cat([torch.from_numpy(s).float(), torch.zeros(1)])
        self.timesteps = torch.from_numpy(ts)
        self.idx = 0

    def scale_model_input(self, sample, t=None):
        sigma = self.sigmas[self.idx]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, t, sample):
        out = ops_ref.euler_step_ref(model_output, sample, self.sigmas[self.idx], self.sigmas[self.idx + 1], self.pred)
        self.idx += 1
        return out


def generate_image_ref(policy: Policy, image_tokens: List[torch.Tensor], codebook: torch.Tensor,
                       decoder_sd: Dict[str, torch.Tensor], unet_sd: Dict[str, torch.Tensor], unet_cfg: dict,
                       scheduler, noise: torch.Tensor, num_return_images=1, width=512, height=512,
                       num_inference_steps=25, guidance_scale=7.0, decoder_heads=16, vae_sd=None, vae_cfg=None,
                       original_size=None, crops_coords_top_left=None):
    """lavit_for_pigeon.py:706-782 with explicit `noise` in place of torch.randn(generator=...) (:716-718).
    Returns dict(latents_per_step, latents, prompt_embeds, pooled, images uint8 NHWC | None)."""
    cdt = policy.compute
    # :709 / :897-920
    emb, token_nums = ops_ref.token_gather_ref(image_tokens, codebook, cdt)
    prompt_embeds, pooled = models_ref.token_decoder_forward(policy, decoder_sd, emb, token_nums, decoder_heads)
    neg_prompt, neg_pooled = torch.zeros_like(prompt_embeds), torch.zeros_like(pooled)          # :710-711
    batch_size = len(prompt_embeds) * num_return_images                                         # :713
    latents = noise.float() * scheduler.init_noise_sigma                                        # :716-719
    assert latents.shape == (batch_size, unet_cfg["in_channels"], height // 8, width // 8)
    scheduler.set_timesteps(num_inference_steps)                                                # :720
    target_size = (width, height)                                                               # :724
    original_size = original_size if original_size is not None else target_size
    crops = crops_coords_top_left if crops_coords_top_left is not None else (0, 0)
    add_time_ids = torch.tensor([list(original_size + crops + target_size)], dtype=prompt_embeds.dtype)  # :791-794
    add_time_ids = add_time_ids.repeat(batch_size, 1)
    input_embeds = torch.cat([neg_prompt, prompt_embeds], dim=0)                                # :731
    add_text = torch.cat([neg_pooled, pooled], dim=0)
    add_time_ids = torch.cat([add_time_ids, add_time_ids], dim=0)
    input_embeds = input_embeds.repeat_interleave(num_return_images, dim=0)                     # :735-736
    add_text = add_text.repeat_interleave(num_return_images, dim=0)
    assert input_embeds.shape[0] == add_text.shape[0] == add_time_ids.shape[0] == latents.shape[0] * 2   # :742
    per_step = []
    for t in scheduler.timesteps:                                                               # :744
        x = scheduler.scale_model_input(torch.cat([latents] * 2), t)                            # :747-748
        noise_pred = models_ref.unet_forward(policy, unet_sd, unet_cfg, x, t, input_embeds,
                                             {"text_embeds": add_text, "time_ids": add_time_ids})  # :752-757
        guided = ops_ref.cfg_combine_ref(noise_pred, guidance_scale)                            # :760-761
        latents = scheduler.step(guided, t, latents)[0]                                         # :764
        per_step.append(latents.clone())
    images = None
    if vae_sd is not None:
        vdt = vae_sd["post_quant_conv.weight"].dtype                                            # :773
        img = models_ref.vae_decode(Policy(vdt), vae_sd, vae_cfg, latents.to(vdt) / vae_cfg["scaling_factor"])  # :774
        img = (img.float() / 2 + 0.5).clamp(0, 1)                                               # :780-781
        img = img.permute(0, 2, 3, 1).numpy()                                                   # :782
        images = (img * 255).round().astype("uint8")                                            # :929
    return {"latents_per_step": per_step, "latents": latents, "prompt_embeds": prompt_embeds, "pooled": pooled,
            "images": images}


def pixel_decoding_origin_ref(policy: Policy, image_tokens: List[torch.Tensor], codebook: torch.Tensor,
                              decoder_sd: Dict[str, torch.Tensor], unet_sd: Dict[str, torch.Tensor], unet_cfg: dict,
                              scheduler, uncond_embeddings: torch.Tensor, noise: torch.Tensor, width=512, height=512,
                              num_inference_steps=50, guidance_scale=5.0, decoder_heads=16, vae_sd=None, vae_cfg=None):
    """lavit_for_pigeon.py:822-895 (the SD-1.5 `lowres` decoder) with explicit `noise` in place of the CPU
    torch.randn(generator=...) of :858-860.  `uncond_embeddings` is the `[1, n_u, 768]` buffer of :168."""
    cdt = policy.compute
    emb, token_nums = ops_ref.token_gather_ref(image_tokens, codebook, cdt)                     # :837 / :897-920
    xrec = models_ref.token_decoder_forward(policy, decoder_sd, emb, token_nums, decoder_heads)
    batch_size = len(xrec)                                                                      # :838
    _, num_tokens, C = xrec.shape
    hidden_uncond = torch.zeros(batch_size, num_tokens, C, dtype=xrec.dtype)                    # :843
    unc = uncond_embeddings[0].to(xrec.dtype)                                                   # :844
    hidden_uncond[:, :len(unc)] = unc                                                           # :845
    encoder_mask = torch.ones(batch_size, num_tokens, dtype=torch.long)                         # :848
    uncond_mask = torch.zeros(batch_size, num_tokens, dtype=torch.long)
    uncond_mask[:, :len(unc)] = 1                                                               # :850
    text_embeddings = torch.cat([hidden_uncond, xrec])                                          # :855
    text_mask = torch.cat([uncond_mask.bool(), encoder_mask.bool()])                            # :856
    latents = noise.float() * scheduler.init_noise_sigma                                        # :858-862
    assert latents.shape == (batch_size, unet_cfg["in_channels"], height // 8, width // 8)
    scheduler.set_timesteps(num_inference_steps)                                                # :864
    per_step = []
    for t in scheduler.timesteps:                                                               # :866
        x = scheduler.scale_model_input(torch.cat([latents] * 2), t)                            # :868-870
        noise_pred = models_ref.unet_forward(policy, unet_sd, unet_cfg, x, t, text_embeddings,
                                             encoder_attention_mask=text_mask)                  # :873-874
        guided = ops_ref.cfg_combine_ref(noise_pred, guidance_scale)                            # :877-878
        latents = scheduler.step(guided, t, latents)[0]                                         # :881
        per_step.append(latents.clone())
    images = None
    if vae_sd is not None:
        # :883-884 — inside the autocast region: the division is fp32, the decoder computes in the compute dtype
        img = models_ref.vae_decode(policy, vae_sd, vae_cfg, latents / vae_cfg["scaling_factor"])
        img = (img.float() / 2 + 0.5).clamp(0, 1)                                               # :886-887
        images = (img.permute(0, 2, 3, 1).numpy() * 255).round().astype("uint8")                # :888, :929
    return {"latents_per_step": per_step, "latents": latents, "xrec": xrec, "images": images}
