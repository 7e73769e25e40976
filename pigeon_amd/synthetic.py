"""Random-init decoder stacks and synthetic token batches (there is no network in the build: no LaVIT checkpoint).
Initialisation happens directly on the GPU from a seeded device generator: 2.57 B UNet parameters in seconds."""
import math
from typing import List, Optional

import torch

from .pipeline import PigeonPixelDecoder
from .schedulers import DDIMScheduler
from .token_decoder import HighresVQDecoder
from .unet import SDXL_CONFIG, UNet2DConditionModel
from .vae import AutoencoderKL

# SDXL-style scheduler values; the LaVIT scheduler_config.json is not shipped with the reference
SDXL_SCHEDULER = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                      steps_offset=1, clip_sample=False, set_alpha_to_one=False, timestep_spacing="leading",
                      prediction_type="epsilon")


def _materialize(module: torch.nn.Module, device, dtype, seed: int):
    module.to_empty(device=device)
    gen = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if leaf in ("pos_embed", "query_embed"):
                p.copy_(torch.randn(p.shape, generator=gen, device=device) * 0.02)
            elif leaf == "positional_embedding":
                p.copy_(torch.randn(p.shape, generator=gen, device=device) / p.shape[-1] ** 0.5)
            elif leaf == "bias":
                p.zero_()
            elif p.dim() == 1:
                p.fill_(1.0)
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=gen, device=device) * (1.0 / fan_in) ** 0.5)
    return module.to(dtype)


def build_random_stack(device="cuda", model_dtype="bf16", unet_config: Optional[dict] = None, decoder_depth: int = 12,
                       vae_config: Optional[dict] = None, with_vae: bool = True, seed: int = 123,
                       scheduler=None) -> PigeonPixelDecoder:
    dtype = torch.bfloat16 if model_dtype == "bf16" else torch.float16
    with torch.device("meta"):
        unet = UNet2DConditionModel(**(unet_config or SDXL_CONFIG))
        dec = HighresVQDecoder(depth=decoder_depth)
        vae = AutoencoderKL(**(vae_config or {})) if with_vae else None
    unet = _materialize(unet, device, dtype, seed)
    dec = _materialize(dec, device, dtype, seed + 1)
    if vae is not None:
        vae = _materialize(vae, device, dtype, seed + 2)
    gen = torch.Generator(device=device).manual_seed(seed + 3)
    codebook = torch.nn.functional.normalize(torch.randn(16384, 32, generator=gen, device=device), p=2, dim=-1)
    return PigeonPixelDecoder(codebook, dec, unet, vae, scheduler or DDIMScheduler(**SDXL_SCHEDULER), model_dtype, device)


def synthetic_tokens(batch: int, seed: int = 123, min_len: int = 20, max_len: int = 256) -> List[torch.Tensor]:
    """n ~ U{20..256} codebook ids + eos 32001 + pad id 2 to the batch max (inference.py:268-282)."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(min_len, max_len + 1, (batch,), generator=g).tolist()
    width = max(lens) + 1
    out = []
    for n in lens:
        ids = torch.randint(32002, 32002 + 16384, (n,), generator=g)
        out.append(torch.cat([ids, torch.tensor([32001]), torch.full((width - n - 1,), 2, dtype=torch.long)]))
    return out
