"""AutoencoderKL decoder half with the diffusers surface the reference uses
(Pigeon/models/lavit_for_pigeon.py:187-188,767-778,796-813): `.decode(z).sample`, `.config.scaling_factor`,
`.config.force_upcast`, `.dtype`, `.post_quant_conv`, `.decoder.mid_block.attentions[0].processor`.

SURVEY.md §8(f) rank 1 ("next"): the ~30 GroupNorm(+SiLU) already run in the HIP kernel; the single 512-wide
attention head of the mid block is outside the d=64 MFMA kernel and uses torch's fused SDPA on the GPU (the reference
leaves it on diffusers' default processor as well: only `self.unet` is switched to xformers at :202).
State-dict names are diffusers'; encoder / quant_conv weights in a checkpoint are ignored (decode-only path).
"""
import json
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from .attention_processor import Attention, WideHeadAttnProcessor
from .layers import GroupNormAct, ResnetBlock2D, Upsample2D, conv2d_bias


class _Config(dict):
    __getattr__ = dict.__getitem__


@dataclass
class DecoderOutput:
    sample: torch.Tensor


SDXL_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       latent_channels=4, norm_num_groups=32, scaling_factor=0.13025, force_upcast=True,
                       act_fn="silu", sample_size=1024)


class _MidBlock(nn.Module):
    def __init__(self, ch, groups, attention_head_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, groups, 1e-6), ResnetBlock2D(ch, ch, None, groups, 1e-6)])
        self.attentions = nn.ModuleList([
            Attention(ch, None, heads=ch // attention_head_dim, dim_head=attention_head_dim, bias=True,
                      norm_num_groups=groups, eps=1e-6, residual_connection=True,
                      processor=WideHeadAttnProcessor())])

    def forward(self, x):
        x = self.resnets[0](x)
        x = self.attentions[0](x)
        return self.resnets[1](x)


class _UpDecoderBlock(nn.Module):
    def __init__(self, in_ch, out_ch, layers, groups, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, None, groups, 1e-6) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Decoder(nn.Module):
    def __init__(self, latent_channels, out_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _MidBlock(rev[0], groups, rev[0])
        self.up_blocks = nn.ModuleList()
        out_ch = rev[0]
        for i, ch in enumerate(rev):
            in_ch, out_ch = out_ch, ch
            self.up_blocks.append(_UpDecoderBlock(in_ch, out_ch, layers_per_block + 1, groups, i != len(rev) - 1))
        self.conv_norm_out = GroupNormAct(groups, rev[-1], eps=1e-6, act=True)
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(conv2d_bias(self.conv_in, z))
        # diffusers Decoder.forward: `sample = sample.to(upscale_dtype)` — after upcast_vae (lavit_for_pigeon.py:796-813)
        # conv_in / mid_block stay in fp16 while the up blocks run in fp32
        x = x.to(next(iter(self.up_blocks.parameters())).dtype)
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(self.conv_norm_out(x))


class AutoencoderKL(nn.Module):
    def __init__(self, **config):
        super().__init__()
        full = dict(SDXL_VAE_CONFIG)
        full.update({k: v for k, v in config.items() if not k.startswith("_")})
        cfg = self.config = _Config(full)
        self.decoder = Decoder(cfg.latent_channels, cfg.out_channels, tuple(cfg.block_out_channels),
                               cfg.layers_per_block, cfg.norm_num_groups)
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **unused):
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "config.json")) as f:
            cfg = json.load(f)
        model = cls(**cfg)
        st = os.path.join(d, "diffusion_pytorch_model.safetensors")
        pt = os.path.join(d, "diffusion_pytorch_model.bin")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(pt, map_location="cpu", weights_only=True)
        renames = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}  # pre-0.18 attention names
        out = {}
        for k, v in sd.items():
            if k.startswith(("encoder.", "quant_conv.")):
                continue
            parts = k.split(".")
            if "attentions" in parts and parts[-2] in renames:
                parts[-2:-1] = renames[parts[-2]].split(".")
                k = ".".join(parts)
            out[k] = v
        model.load_state_dict(out)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    def enable_channels_last(self, enabled: bool = True):
        """See UNet2DConditionModel.enable_channels_last."""
        self.channels_last = enabled
        self.to(memory_format=torch.channels_last if enabled else torch.contiguous_format)
        return self

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        if getattr(self, "channels_last", False):
            z = z.contiguous(memory_format=torch.channels_last)
        dec = self.decoder(self.post_quant_conv(z))
        return DecoderOutput(sample=dec) if return_dict else (dec,)
