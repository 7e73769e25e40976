"""Building blocks shared by the UNet and the VAE decoder (state-dict names follow diffusers 0.27.2)."""
import math
import os
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


def _conv_nobias(conv: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


# A/B switch for measurements (scripts/): PGN_FUSE_CONV_BIAS=0 leaves every bias / residual add to torch
FUSE_CONV_BIAS = os.environ.get("PGN_FUSE_CONV_BIAS", "1") != "0"


def _bias_fusable(conv: nn.Conv2d, x: torch.Tensor) -> bool:
    """MIOpen convolutions have no bias epilogue: torch adds the bias in a second, strided-broadcast launch.  When the
    output shape allows 8-element vectors (always, except the 3/4-channel conv_out layers) the HIP kernel does it."""
    return (FUSE_CONV_BIAS and conv.bias is not None and x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16)
            and conv.out_channels % 8 == 0)


# PGN_PRODUCER_STATS=0: every GroupNorm computes its own statistics (A/B switch).  Default: the launch that adds a
# convolution's bias / a block's residual also emits the partial statistics of the GroupNorm that reads its output
# (ops.bias_residual_add(..., gn_stats=True)): that GroupNorm then reads the tensor once instead of twice.
PRODUCER_STATS = os.environ.get("PGN_PRODUCER_STATS", "1") != "0"


def conv2d_bias(conv: nn.Conv2d, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv(x) [+ residual] with the bias (and the residual) added by ops.bias_residual_add in one launch."""
    if not _bias_fusable(conv, x):
        y = conv(x)
        return y if residual is None else y + residual
    y = _conv_nobias(conv, x)
    if not (y.is_contiguous(memory_format=torch.channels_last) or (y.is_contiguous() and (y.shape[2] * y.shape[3]) % 8 == 0)):
        y = y + conv.bias.view(1, -1, 1, 1)
        return y if residual is None else y + residual
    return ops.bias_residual_add(y, conv.bias, residual, gn_stats=PRODUCER_STATS)


# PGN_SPLIT_CONCAT=0: the up blocks write torch.cat([hidden, skip]) and the 1x1 shortcuts stay MIOpen convolutions
SPLIT_CONCAT = os.environ.get("PGN_SPLIT_CONCAT", "1") != "0"


# PGN_KEEP_DERIVED=0: recompute the derived bias vectors / the time embedding's SiLU in every forward (A/B switch)
KEEP_DERIVED = os.environ.get("PGN_KEEP_DERIVED", "1") != "0"


def derived_constant(owner, slot: str, sources, build):
    """A tensor computed from frozen parameters (an fp32 copy of a bias, the sum of two biases, ...) kept on `owner`
    instead of being recomputed by two to four tiny launches in every forward (at batch 1 a UNet forward is ~3 k
    launches, and ~120 of them were these).  Keyed like BasicTransformerBlock._pending_biases: storage, dtype and
    VERSION of every source, so an in-place update (load_state_dict, a LoRA merge) rebuilds it."""
    if not KEEP_DERIVED:
        with torch.no_grad():
            return build()
    key = tuple(None if t is None else (t.data_ptr(), t.dtype, t._version) for t in sources)
    hit = owner.__dict__.get(slot)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, build())
        owner.__dict__[slot] = hit
    return hit[1]


def silu_once(temb: torch.Tensor) -> torch.Tensor:
    """F.silu(temb) for the resnets' time_emb_proj: every ResnetBlock2D of a forward applies it to the SAME embedding
    (17 launches in the SDXL UNet); the first caller's result rides on the tensor object."""
    hit = getattr(temb, "_pgn_silu", None) if KEEP_DERIVED else None
    if hit is not None and hit[0] == temb._version:
        return hit[1]
    s = F.silu(temb)
    if not (torch.is_grad_enabled() and temb.requires_grad):
        temb._pgn_silu = (temb._version, s)
    return s


def _nhwc_rows(x: torch.Tensor) -> torch.Tensor:
    """[N*H*W, C] view of a channels_last activation."""
    return x.permute(0, 2, 3, 1).view(-1, x.shape[1])


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

    def _gemm_shortcut_ok(self, x: torch.Tensor, skip: Optional[torch.Tensor]) -> bool:
        sc = self.conv_shortcut
        ok = (SPLIT_CONCAT and sc is not None and sc.kernel_size == (1, 1) and sc.stride == (1, 1) and sc.groups == 1
              and sc.padding == (0, 0) and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16)
              and sc.weight.dtype == x.dtype and ops._is_channels_last(x) and _bias_fusable(self.conv2, x)
              and x.shape[1] % 8 == 0 and x.shape[1] // self.norm1.num_groups <= 128)
        if ok and skip is not None:
            c = x.shape[1] + skip.shape[1]
            ok = (ops._is_channels_last(skip) and skip.dtype == x.dtype and skip.shape[1] % 8 == 0
                  and c == self.norm1.num_channels and c // self.norm1.num_groups <= 128 and c <= 8192)
        return ok

    def _shortcut_bias(self, dtype):
        """conv2's bias + the shortcut's, summed in fp32 and rounded once (one vector for pgn_bias_residual_add)."""
        b2, bs = self.conv2.bias, self.conv_shortcut.bias
        return derived_constant(self, f"_pgn_b2s_{dtype}", (b2, bs),
                                lambda: (b2.detach().float() + bs.detach().float()).to(dtype))

    def forward(self, input_tensor, temb=None, skip=None):
        """diffusers ResnetBlock2D.forward.  Three broadcast / tensor adds never run as launches of their own:
        conv1's bias and the time-embedding projection travel as ONE per-(sample, channel) vector into norm2's
        kernel; conv2's bias (+ the shortcut's) and the residual add are one pgn_bias_residual_add launch.

        skip: the up blocks' `torch.cat([hidden_states, res_hidden_states], dim=1)` handed over as its two halves
        (input_tensor, skip).  On the GPU in channels_last the concatenation is never written: norm1 reads both
        halves in place (ops.group_norm_silu_cat) and the 1x1 shortcut is two GEMMs over the halves, accumulated
        (beta = 1) into conv2's output; a 1x1 shortcut on a single input is one such GEMM."""
        gemm_sc = self._gemm_shortcut_ok(input_tensor, skip)
        if skip is not None and not gemm_sc:
            input_tensor, skip = torch.cat([input_tensor, skip], dim=1), None
        if skip is not None:
            x = ops.group_norm_silu_cat(input_tensor, skip, self.norm1.num_groups, self.norm1.weight, self.norm1.bias,
                                        self.norm1.eps, silu=self.norm1.act)
        else:
            x = self.norm1(input_tensor)
        t = None
        if self.time_emb_proj is not None and temb is not None:
            t = self.time_emb_proj(silu_once(temb))   # [N, C]: its broadcast add is fused into norm2's kernel
        if t is not None and _bias_fusable(self.conv1, x):
            h = _conv_nobias(self.conv1, x)
            b1 = derived_constant(self, "_pgn_b1_f32", (self.conv1.bias,), lambda: self.conv1.bias.detach().float())
            t = torch.add(t, b1)                   # = t.float() + conv1.bias.float() (type promotion), one launch
        else:
            h = conv2d_bias(self.conv1, x)
        h = self.norm2(h, add_nc=t)
        if self.conv_shortcut is None:
            out = conv2d_bias(self.conv2, h, input_tensor)
        elif gemm_sc:
            hh = _conv_nobias(self.conv2, h)
            if not ops._is_channels_last(hh):
                hh = hh.contiguous(memory_format=torch.channels_last)
            w = self.conv_shortcut.weight.flatten(1)           # [C_out, C_in]
            c1 = input_tensor.shape[1]
            rows = _nhwc_rows(hh)
            rows.addmm_(_nhwc_rows(input_tensor), w[:, :c1].t())
            if skip is not None:
                rows.addmm_(_nhwc_rows(skip), w[:, c1:].t())
            b = self._shortcut_bias(hh.dtype)
            out = ops.bias_residual_add(hh, b, None, gn_stats=PRODUCER_STATS)
        elif _bias_fusable(self.conv2, h):
            # both biases in one vector: shortcut(x) + b_s + conv2(h) + b_2
            sc = _conv_nobias(self.conv_shortcut, input_tensor)
            hh = _conv_nobias(self.conv2, h)
            if hh.stride() == sc.stride() and (hh.is_contiguous(memory_format=torch.channels_last)
                                               or (hh.is_contiguous() and (hh.shape[2] * hh.shape[3]) % 8 == 0)):
                b = self._shortcut_bias(hh.dtype)
                out = ops.bias_residual_add(hh, b, sc, gn_stats=PRODUCER_STATS)
            else:
                out = (sc + self.conv_shortcut.bias.view(1, -1, 1, 1)) + (hh + self.conv2.bias.view(1, -1, 1, 1))
        else:
            out = self.conv_shortcut(input_tensor) + self.conv2(h)
        return out if self.output_scale_factor == 1.0 else out / self.output_scale_factor


class Downsample2D(nn.Module):
    def __init__(self, channels: int, padding: int = 1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return conv2d_bias(self.conv, x)


HIP_UPSAMPLE = os.environ.get("PGN_HIP_UPSAMPLE", "1") != "0"


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None or tuple(output_size) == (2 * x.shape[2], 2 * x.shape[3]):
            if (HIP_UPSAMPLE and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.shape[1] % 8 == 0
                    and x.numel() and ops._is_channels_last(x)):
                x = ops.upsample_nearest2x(x)      # torch's channels_last kernel runs at 0.7 TB/s
            else:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return conv2d_bias(self.conv, x)
