"""Building blocks shared by the UNet and the VAE decoder (state-dict names follow diffusers 0.27.2)."""
import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class GroupNormAct(nn.Module):
    """nn.GroupNorm-compatible parameters (`weight`, `bias`), evaluated by the fused HIP GroupNorm(+SiLU) kernel.
    Replaces the `nonlinearity(norm(x))` pairs of diffusers ResnetBlock2D / conv_norm_out / Transformer2DModel.norm."""

    def __init__(self, num_groups: int, num_channels: int, eps: float = 1e-5, affine: bool = True, act: bool = True):
        super().__init__()
        self.num_groups, self.num_channels, self.eps, self.act = num_groups, num_channels, eps, act
        if affine:
            self.weight = nn.Parameter(torch.ones(num_channels))
            self.bias = nn.Parameter(torch.zeros(num_channels))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def forward(self, x: torch.Tensor, add_nc: Optional[torch.Tensor] = None) -> torch.Tensor:
        return ops.group_norm_silu(x, self.num_groups, self.weight, self.bias, self.eps, silu=self.act, add_nc=add_nc)


def timestep_embedding(timesteps: torch.Tensor, dim: int, flip_sin_to_cos: bool, downscale_freq_shift: float,
                       max_period: int = 10000) -> torch.Tensor:
    """diffusers get_timestep_embedding (sinusoidal), fp32."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(F.silu(self.linear_1(sample)))


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: Optional[int], groups: int = 32,
                 eps: float = 1e-5, output_scale_factor: float = 1.0):
        super().__init__()
        self.norm1 = GroupNormAct(groups, in_channels, eps=eps, act=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = GroupNormAct(groups, out_channels, eps=eps, act=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = output_scale_factor

    def forward(self, input_tensor, temb=None):
        h = self.conv1(self.norm1(input_tensor))
        t = None
        if self.time_emb_proj is not None and temb is not None:
            t = self.time_emb_proj(F.silu(temb))   # [N, C]: its broadcast add is fused into norm2's kernel
        h = self.conv2(self.norm2(h, add_nc=t))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        out = input_tensor + h
        return out if self.output_scale_factor == 1.0 else out / self.output_scale_factor


class Downsample2D(nn.Module):
    def __init__(self, channels: int, padding: int = 1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)
