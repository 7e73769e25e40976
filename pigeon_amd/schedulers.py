"""Schedulers with the diffusers 0.27.2 object protocol the reference drives
(Pigeon/models/lavit_for_pigeon.py:189,719-720,748,764): `.init_noise_sigma`, `.set_timesteps(n, device=)`,
`.timesteps`, `.scale_model_input(x, timestep=t)`, `.step(noise_pred, t, latents).prev_sample`.

MI355X-first differences (results unchanged):
  * every per-step coefficient is precomputed on the HOST when `set_timesteps` runs, with the same fp32 tensor
    arithmetic diffusers uses, so `step` never indexes `alphas_cumprod` with a device timestep (the reference pays
    one host<->device sync per step for that);
  * `step` is ONE launch of the fused HIP kernel (pigeon_amd.ops.cfg_ddim_step / cfg_euler_step);
  * `step_cfg(noise_pred_2B, t, latents, guidance_scale)` additionally folds `chunk(2)` + the guidance
    combine (lavit_for_pigeon.py:760-761) into that launch.
"""
import json
import math
import os
from dataclasses import dataclass
from typing import Optional, Union

import numpy as np
import torch

from . import ops


class _Config(dict):
    __getattr__ = dict.__getitem__


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


def _betas(beta_schedule, beta_start, beta_end, num_train_timesteps, trained_betas):
    if trained_betas is not None:
        return torch.tensor(trained_betas, dtype=torch.float32)
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    if beta_schedule == "squaredcos_cap_v2":
        def bar(t):
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        b = [min(1 - bar((i + 1) / num_train_timesteps) / bar(i / num_train_timesteps), 0.999)
             for i in range(num_train_timesteps)]
        return torch.tensor(b, dtype=torch.float32)
    raise NotImplementedError(f"{beta_schedule} is not implemented")


def _rescale_zero_terminal_snr(betas):
    alphas_bar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    first, last = alphas_bar_sqrt[0].clone(), alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt = (alphas_bar_sqrt - last) * first / (first - last)
    alphas_bar = alphas_bar_sqrt ** 2
    alphas = torch.cat([alphas_bar[0:1], alphas_bar[1:] / alphas_bar[:-1]])
    return 1 - alphas


def _load_config(path, subfolder):
    d = os.path.join(path, subfolder) if subfolder else path
    with open(os.path.join(d, "scheduler_config.json")) as f:
        cfg = json.load(f)
    return {k: v for k, v in cfg.items() if not k.startswith("_")}


def _timestep_key(t) -> int:
    return int(t.item()) if torch.is_tensor(t) else int(t)


class DDIMScheduler:
    """DDIM (eta = 0) with the fused CFG+step kernel.  Defaults are diffusers' DDIMScheduler defaults."""

    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, **unused):
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
                              set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                              prediction_type=prediction_type, thresholding=thresholding,
                              dynamic_thresholding_ratio=dynamic_thresholding_ratio,
                              clip_sample_range=clip_sample_range, sample_max_value=sample_max_value,
                              timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr)
        self.betas = _betas(beta_schedule, beta_start, beta_end, num_train_timesteps, trained_betas)
        if rescale_betas_zero_snr:
            self.betas = _rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self._coef = {}
        self._host_timesteps = self.timesteps.tolist()

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **overrides):
        cfg = _load_config(path, subfolder)
        cfg.update(overrides)
        return cls(**cfg)

    @classmethod
    def from_config(cls, config, **overrides):
        cfg = dict(config)
        cfg.update(overrides)
        return cls(**cfg)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def scale_divisor(self, timestep=None) -> float:
        """Divisor `scale_model_input` applies (1 for DDIM); lets the pipeline fuse it into cfg_expand_scale."""
        return 1.0

    def set_timesteps(self, num_inference_steps: int, device: Union[str, torch.device, None] = None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {T}")
        self.num_inference_steps = num_inference_steps
        spacing = self.config.timestep_spacing
        if spacing == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif spacing == "leading":
            ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif spacing == "trailing":
            ratio = T / num_inference_steps
            ts = np.round(np.arange(T, 0, -ratio)).astype(np.int64)
            ts -= 1
        else:
            raise ValueError(f"{spacing} is not supported. Choose one of 'leading', 'trailing' or 'linspace'.")
        self._host_timesteps = [int(t) for t in ts]
        self.timesteps = torch.from_numpy(ts).to(device)
        # per-step fp32 coefficients, computed as diffusers computes them inside step()
        self._coef = {}
        for t in self._host_timesteps:
            prev_t = t - T // num_inference_steps
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
            b_t = 1 - a_t
            std_dev_t = 0.0
            self._coef[t] = (float(a_t ** 0.5), float(b_t ** 0.5), float(a_prev ** 0.5),
                             float((1 - a_prev - std_dev_t ** 2) ** 0.5))

    # -- thresholding runs between two launches (quantile needs a global reduction) -----------------
    def _threshold_sample(self, sample):
        dtype = sample.dtype
        b, c, *rest = sample.shape
        if dtype not in (torch.float32, torch.float64):
            sample = sample.float()
        sample = sample.reshape(b, c * int(np.prod(rest)))
        s = torch.quantile(sample.abs(), self.config.dynamic_thresholding_ratio, dim=1)
        s = torch.clamp(s, min=1, max=self.config.sample_max_value).unsqueeze(1)
        sample = torch.clamp(sample, -s, s) / s
        return sample.reshape(b, c, *rest).to(dtype)

    def _step(self, eps_uncond, eps_text, guidance_scale, timestep, sample, use_clipped_model_output, eta):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating "
                             "the scheduler")
        if eta != 0.0:
            raise NotImplementedError("pigeon_amd DDIMScheduler implements the deterministic eta=0 step the reference uses")
        t = _timestep_key(timestep)
        if t not in self._coef:
            raise ValueError(f"timestep {t} is not in the current schedule")
        sa_t, sb_t, sa_prev, dirc = self._coef[t]
        pred = self.config.prediction_type
        if pred not in ("epsilon", "sample", "v_prediction"):
            raise ValueError(f"prediction_type given as {pred} must be one of `epsilon`, `sample`, or `v_prediction`")
        if self.config.thresholding:
            # launch 1: x0 only; torch quantile/clamp; launch 2: finish from the thresholded x0 ("sample" form)
            _, x0 = ops.cfg_ddim_step(None, sample, guidance_scale, sa_t, sb_t, sa_prev, dirc, prediction_type=pred,
                                      eps_uncond=eps_uncond, eps_text=eps_text, return_pred_x0=True)
            x0 = self._threshold_sample(x0)
            if use_clipped_model_output or pred == "sample":
                prev, _ = ops.cfg_ddim_step(None, sample, 0.0, sa_t, sb_t, sa_prev, dirc, prediction_type="sample",
                                            eps_text=x0)
            else:
                raise NotImplementedError("thresholding without use_clipped_model_output is not fused; "
                                          "set use_clipped_model_output=True")
            return SchedulerOutput(prev, x0)
        clip = self.config.clip_sample_range if self.config.clip_sample else None
        prev, x0 = ops.cfg_ddim_step(None, sample, guidance_scale, sa_t, sb_t, sa_prev, dirc, prediction_type=pred,
                                     eps_uncond=eps_uncond, eps_text=eps_text, clip_sample_range=clip,
                                     use_clipped_model_output=use_clipped_model_output, return_pred_x0=False)
        return SchedulerOutput(prev, x0)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True):
        """diffusers signature; model_output is the already-guided noise prediction."""
        out = self._step(None, model_output, 0.0, timestep, sample, use_clipped_model_output, eta)
        return out if return_dict else (out.prev_sample,)

    def step_cfg(self, noise_pred, timestep, sample, guidance_scale: float, use_clipped_model_output: bool = False):
        """Fused lavit_for_pigeon.py:760-764: noise_pred is the UNet output for [uncond; text]."""
        half = noise_pred.shape[0] // 2
        noise_pred = noise_pred.contiguous()
        return self._step(noise_pred[:half], noise_pred[half:], guidance_scale, timestep, sample,
                          use_clipped_model_output, 0.0)

    def __len__(self):
        return self.config.num_train_timesteps


class EulerDiscreteScheduler:
    """Euler (ancestral-free, s_churn = 0) scheduler, diffusers 0.27.2 semantics, fused CFG+step kernel."""

    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, prediction_type="epsilon", interpolation_type="linear", use_karras_sigmas=False,
                 sigma_min=None, sigma_max=None, timestep_spacing="linspace", timestep_type="discrete",
                 steps_offset=0, rescale_betas_zero_snr=False, keep_fp32_latents=False, **unused):
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, trained_betas=trained_betas,
                              prediction_type=prediction_type, interpolation_type=interpolation_type,
                              use_karras_sigmas=use_karras_sigmas, sigma_min=sigma_min, sigma_max=sigma_max,
                              timestep_spacing=timestep_spacing, timestep_type=timestep_type,
                              steps_offset=steps_offset, rescale_betas_zero_snr=rescale_betas_zero_snr)
        if use_karras_sigmas or interpolation_type != "linear" or timestep_type != "discrete":
            raise NotImplementedError("only linear-interpolated discrete sigmas are implemented")
        self.betas = _betas(beta_schedule, beta_start, beta_end, num_train_timesteps, trained_betas)
        if rescale_betas_zero_snr:
            self.betas = _rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        if rescale_betas_zero_snr:
            self.alphas_cumprod[-1] = 2 ** -24
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        timesteps = np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].copy()
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.float32)
        self.sigmas = torch.cat([torch.from_numpy(sigmas[::-1].copy()).to(dtype=torch.float32), torch.zeros(1)])
        self.num_inference_steps = None
        self.keep_fp32_latents = keep_fp32_latents  # non-reference option: do not cast prev_sample back to bf16
        self._step_index = None
        self._host_timesteps = [float(t) for t in timesteps]

    from_pretrained = classmethod(DDIMScheduler.from_pretrained.__func__)
    from_config = classmethod(DDIMScheduler.from_config.__func__)

    @property
    def init_noise_sigma(self):
        max_sigma = self.sigmas.max()
        if self.config.timestep_spacing in ("linspace", "trailing"):
            return max_sigma
        return (max_sigma ** 2 + 1) ** 0.5

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        spacing = self.config.timestep_spacing
        if spacing == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        elif spacing == "leading":
            ratio = T // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.float32)
            ts += self.config.steps_offset
        elif spacing == "trailing":
            ratio = T / num_inference_steps
            ts = (np.arange(T, 0, -ratio)).round().copy().astype(np.float32)
            ts -= 1
        else:
            raise ValueError(f"{spacing} is not supported. Choose one of 'linspace', 'leading' or 'trailing'.")
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.interp(ts, np.arange(0, len(sigmas)), sigmas)
        sigmas = torch.from_numpy(sigmas).to(dtype=torch.float32)
        self.timesteps = torch.from_numpy(ts.astype(np.float32)).to(device=device)
        self._host_timesteps = [float(t) for t in ts]
        self.sigmas = torch.cat([sigmas, torch.zeros(1)])  # kept on the host: no device sync per step
        self._step_index = None

    def _init_step_index(self, timestep):
        t = float(timestep.item()) if torch.is_tensor(timestep) else float(timestep)
        idx = [i for i, v in enumerate(self._host_timesteps) if v == t]
        if not idx:
            raise ValueError(f"timestep {t} is not in the current schedule")
        # diffusers picks the second match when a timestep is duplicated (image-to-image start); first otherwise
        self._step_index = idx[1] if len(idx) > 1 else idx[0]

    def scale_divisor(self, timestep=None) -> float:
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self._step_index]
        return float((sigma ** 2 + 1) ** 0.5)

    def scale_model_input(self, sample, timestep=None):
        return sample / self.scale_divisor(timestep)

    def _step(self, eps_uncond, eps_text, guidance_scale, timestep, sample):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma, sigma_next = self.sigmas[self._step_index], self.sigmas[self._step_index + 1]
        out_dtype = torch.float32 if self.keep_fp32_latents else None
        prev, x0 = ops.cfg_euler_step(None, sample, guidance_scale, float(sigma), float(sigma_next),
                                      self.config.prediction_type, eps_uncond=eps_uncond, eps_text=eps_text,
                                      out_dtype=out_dtype)
        self._step_index += 1
        return SchedulerOutput(prev, x0)

    def step(self, model_output, timestep, sample, s_churn: float = 0.0, s_tmin: float = 0.0,
             s_tmax: float = float("inf"), s_noise: float = 1.0, generator=None, return_dict: bool = True):
        if s_churn != 0.0:
            raise NotImplementedError("s_churn > 0 (stochastic Euler) is not implemented")
        out = self._step(None, model_output, 0.0, timestep, sample)
        return out if return_dict else (out.prev_sample,)

    def step_cfg(self, noise_pred, timestep, sample, guidance_scale: float):
        half = noise_pred.shape[0] // 2
        noise_pred = noise_pred.contiguous()
        return self._step(noise_pred[:half], noise_pred[half:], guidance_scale, timestep, sample)

    def __len__(self):
        return self.config.num_train_timesteps
