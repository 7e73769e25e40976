"""Scoring stage: CLIP ViT-H-14 image tower + per-image scores on the MI355X.

Host-side mirror of what Evaluation/cal_scores.py:144-203 computes for every generated image through
Evaluation/eval_utils.py::CLIPScore (:112-156) — `open_clip.create_model_and_transforms('ViT-H-14',
pretrained="laion2b-s32b-b79K")`, `model.encode_image`, L2 normalisation, `100 * cosine_similarity` against the
target text features (tgt_score) and against the 5 cached history image features (hist_score, mean of 5).

open_clip is a third-party wheel absent from /root/reference and from this image: PARITY UNPINNED.  The tower below
restates open_clip's `VisionTransformer` (state-dict names of its `visual.` subtree, so `open_clip_pytorch_model.bin`
loads after stripping the prefix): conv1 14x14/14 patch embedding, class token, learned positions, ln_pre,
32 pre-norm residual blocks (fused-qkv nn.MultiheadAttention, GELU MLP), ln_post on the class token, `proj`.
On the GPU the 16 x 80-wide attention heads run in the HIP MFMA kernel (pgn_attention_fwd, head dim zero-filled to
128 in-kernel, q/k/v consumed in place from the fused qkv projection), the LayerNorms (+ residual adds) in
pgn_add_layernorm, the scores in pgn_clip_scores; GEMMs on hipBLASLt.  CPU tensors raise (no torch fallback).

The text side (tokenizer + text tower, eval_utils.py:115,126) is pigeon_amd/clip_text.py; `CLIPScorer` takes both
optionally, so target texts can be passed as strings (`texts=`) or, as before, as cached features (`text_features=`,
the way the reference caches `*_hist_embeds.npy`, cal_scores.py:73-99).
"""
import os
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

VIT_H_14 = dict(image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0, output_dim=1024)
# open_clip's OPENAI_DATASET_MEAN / STD (the `clip_trans` of cal_scores.py:110)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _layer_norm(x, residual, ln: nn.LayerNorm):
    """(x + residual, LayerNorm(x + residual)): one pgn_add_layernorm launch; no torch evaluation (CPU tensors raise)."""
    return ops.add_layer_norm(x, residual, ln.weight, ln.bias, ln.eps)


class _MHA(nn.Module):
    """nn.MultiheadAttention parameter layout (in_proj_weight [3C, C], in_proj_bias, out_proj)."""

    def __init__(self, width: int, heads: int):
        super().__init__()
        self.heads = heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)
        nn.init.normal_(self.in_proj_weight, std=width ** -0.5)

    def forward(self, x, causal: bool = False):
        B, N, C = x.shape
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias)          # [B, N, 3C] = [q | k | v]
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        D = C // self.heads
        # image tower: pgn_attention_fwd (head dim 80); text tower: pgn_attention_causal_fwd (77 tokens, head dim 64)
        return self.out_proj(ops.attention(q, k, v, self.heads, D ** -0.5, causal=causal))


class _MLP(nn.Module):
    def __init__(self, width: int, hidden: int):
        super().__init__()
        self.c_fc = nn.Linear(width, hidden)
        self.c_proj = nn.Linear(hidden, width)

    def forward(self, x):
        return self.c_proj(F.gelu(self.c_fc(x)))


class _ResBlock(nn.Module):
    def __init__(self, width: int, heads: int, mlp_ratio: float):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = _MHA(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = _MLP(width, int(width * mlp_ratio))


class _Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, mlp_ratio: float):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResBlock(width, heads, mlp_ratio) for _ in range(layers)])


class CLIPVisionTransformer(nn.Module):
    """open_clip VisionTransformer.forward for the class-token pooled, projected embedding (`encode_image`)."""

    def __init__(self, image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0, output_dim=1024):
        super().__init__()
        self.config = dict(image_size=image_size, patch_size=patch_size, width=width, layers=layers, heads=heads,
                           mlp_ratio=mlp_ratio, output_dim=output_dim)
        grid = image_size // patch_size
        scale = width ** -0.5
        self.conv1 = nn.Conv2d(3, width, patch_size, stride=patch_size, bias=False)
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(grid * grid + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads, mlp_ratio)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    @property
    def dtype(self):
        return self.conv1.weight.dtype

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, **config):
        """`open_clip_pytorch_model.bin` / `.safetensors` of laion/CLIP-ViT-H-14-laion2B-s32B-b79K (what
        cal_scores.py:110 downloads): the `visual.` subtree is kept, everything else ignored."""
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
        model = cls(**{**VIT_H_14, **config})
        model.load_state_dict(sd)
        return (model.to(torch_dtype) if torch_dtype is not None else model).eval()

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values [B, 3, S, S] (normalised) -> image features [B, output_dim] in the parameter dtype."""
        x = self.conv1(pixel_values.to(self.dtype))                       # [B, width, g, g]
        B, W = x.shape[0], x.shape[1]
        x = x.reshape(B, W, -1).permute(0, 2, 1)
        x = torch.cat([self.class_embedding.to(x.dtype).expand(B, 1, W), x], dim=1)
        x = (x + self.positional_embedding.to(x.dtype)).contiguous()
        _, x = _layer_norm(x, None, self.ln_pre)
        pending = None                                                    # mlp output not yet added to x
        for blk in self.transformer.resblocks:
            x, h = _layer_norm(x, pending, blk.ln_1)                      # x += mlp(prev) ; h = ln_1(x)
            a = blk.attn(h)
            x, h = _layer_norm(a, x, blk.ln_2)                            # x += attn ; h = ln_2(x)
            pending = blk.mlp(h)
        x = x[:, 0] + pending[:, 0]                                       # only the class token is read out
        _, pooled = _layer_norm(x.contiguous(), None, self.ln_post)
        return pooled @ self.proj.to(pooled.dtype)


def clip_preprocess(images_u8: torch.Tensor, resize_hw: Optional[Tuple[int, int]] = None, image_size: int = 224,
                    dtype=torch.bfloat16) -> torch.Tensor:
    """uint8 NHWC images on the device -> normalised [M, 3, image_size, image_size].
    Follows cal_scores.py:108-110,153-155: bilinear Resize to `resize_hw` (the run's (height, width)), then open_clip's
    eval transform (bicubic Resize of the short side to 224, centre crop, ToTensor, Normalize).  The reference runs
    these in PIL on JPEG-decoded files; here torch's antialiased interpolation runs on the in-memory images, so inputs
    agree with the reference's to interpolation / JPEG rounding (parity unpinned, see module docstring)."""
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
        raise ValueError("images must be uint8 [M, H, W, 3]")
    x = images_u8.permute(0, 3, 1, 2).float()
    if resize_hw is not None and tuple(x.shape[-2:]) != tuple(resize_hw):
        x = F.interpolate(x, size=tuple(resize_hw), mode="bilinear", antialias=True, align_corners=False)
        x = x.round().clamp(0, 255)                                       # PIL keeps uint8 between the two resizes
    h, w = x.shape[-2:]
    if (h, w) != (image_size, image_size):
        s = image_size / min(h, w)
        nh, nw = max(image_size, round(h * s)), max(image_size, round(w * s))
        x = F.interpolate(x, size=(nh, nw), mode="bicubic", antialias=True, align_corners=False)
        x = x.round().clamp(0, 255)
        top, left = (nh - image_size) // 2, (nw - image_size) // 2
        x = x[:, :, top:top + image_size, left:left + image_size]
    mean = torch.tensor(CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=x.device).view(1, 3, 1, 1)
    return ((x / 255.0 - mean) / std).to(dtype)


class CLIPScorer:
    """Per-image (tgt_score, hist_score) for a batch of generated images, all on one GPU."""

    def __init__(self, vision: CLIPVisionTransformer, device="cuda", model_dtype=torch.bfloat16, batch_size: int = 64,
                 text=None, tokenizer=None):
        """`text` / `tokenizer`: a clip_text.CLIPTextTransformer and SimpleTokenizer (eval_utils.py:114-115 builds both
        halves of one checkpoint); without them target texts must come in as precomputed `text_features`."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("CLIPScorer runs on an MI355X (device 'cuda:N'); there is no CPU path")
        self.vision = vision.to(self.device, model_dtype).eval().requires_grad_(False)
        self.text = None if text is None else text.to(self.device, model_dtype).eval().requires_grad_(False)
        self.tokenizer = tokenizer
        self.batch_size = batch_size                                      # cal_scores.py:47

    @torch.no_grad()
    def encode_text(self, texts: Sequence[str]) -> torch.Tensor:
        """eval_utils.py:126 (`self.model.encode_text(self.tokenizer(texts))`) -> [len(texts), output_dim]."""
        if self.text is None or self.tokenizer is None:
            raise RuntimeError("CLIPScorer was built without a text tower / tokenizer: pass text_features instead")
        if len(texts) == 0:
            return torch.empty(0, self.text.config["output_dim"], device=self.device, dtype=self.text.dtype)
        feats = []
        for i in range(0, len(texts), self.batch_size):
            tokens = self.tokenizer(list(texts[i:i + self.batch_size])).to(self.device)
            feats.append(self.text(tokens))
        return torch.cat(feats, dim=0)

    @torch.no_grad()
    def encode_image(self, images_u8: torch.Tensor, resize_hw=None) -> torch.Tensor:
        feats = []
        for i in range(0, images_u8.shape[0], self.batch_size):
            px = clip_preprocess(images_u8[i:i + self.batch_size].to(self.device), resize_hw,
                                 self.vision.config["image_size"], self.vision.dtype)
            feats.append(self.vision(px))
        return torch.cat(feats, dim=0)

    @torch.no_grad()
    def score(self, images_u8, text_features: Optional[torch.Tensor] = None,
              history_features: Optional[torch.Tensor] = None, resize_hw=None,
              texts: Optional[Sequence[str]] = None) -> torch.Tensor:
        """-> fp32 [M, 2] = (tgt_score, hist_score); a column without its inputs is NaN.  `images_u8`: uint8
        [M, H, W, 3] tensor, or the list of equally sized PIL images generate_image returns.  `texts`: one target
        text per image, encoded here (needs the text tower); alternative to `text_features`."""
        if texts is not None:
            if text_features is not None:
                raise ValueError("pass texts or text_features, not both")
            text_features = self.encode_text(texts)
        if not isinstance(images_u8, torch.Tensor):
            import numpy as np
            images_u8 = torch.from_numpy(np.stack([np.asarray(im.convert("RGB")) for im in images_u8]))
        feats = self.encode_image(images_u8, resize_hw)
        tf = None if text_features is None else text_features.to(self.device)
        hf = None if history_features is None else history_features.to(self.device)
        tgt, hist = ops.clip_scores(feats, tf, hf)
        nan = torch.full((feats.shape[0],), float("nan"), device=self.device)
        return torch.stack([nan if tgt is None else tgt, nan if hist is None else hist], dim=1)
