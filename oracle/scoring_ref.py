"""Scoring-stage oracle (CPU).  TEST INFRASTRUCTURE ONLY.

`clip_scores_ref` follows Evaluation/eval_utils.py:121-129 (CLIPScore.calculate_clip_score), :540-548
(evaluate_personalization_given_data_sim) and the mean over the 5 history images of Evaluation/cal_scores.py:178-179
line by line.  `clip_vision_forward` restates open_clip's VisionTransformer.forward (`model.encode_image`, reached from
eval_utils.py:120,541) over a plain state dict with open_clip's parameter names.  PARITY UNPINNED for the tower:
open_clip is a third-party wheel absent from /root/reference and this image, and the reference holds no fixture for it;
the score arithmetic is pinned by closed-form known answers (tests/test_scoring.py).  `clip_text_forward` restates
open_clip's `CLIP.encode_text` (eval_utils.py:126) the same way, equally unpinned; the BPE tokenizer is cross-checked
against transformers' independent CLIPTokenizer on a synthetic merges table (tests/test_clip_text.py).
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import ops_ref
from .models_ref import SD, Policy


def clip_scores_ref(img: torch.Tensor, txt: Optional[torch.Tensor], hist: Optional[torch.Tensor]):
    """img [M, D], txt [M, D] | None, hist [M, K, D] | None -> (tgt_score [M] | None, hist_score [M] | None), fp32."""
    img = img.float()
    img_n = img / img.norm(p=2, dim=-1, keepdim=True)                                   # eval_utils.py:121 / :542
    tgt = None
    if txt is not None:
        txt = txt.float()
        txt_n = txt / txt.norm(p=2, dim=-1, keepdim=True)                               # :124
        tgt = 100 * F.cosine_similarity(img_n, txt_n)                                   # :126
    hs = None
    if hist is not None:
        hist = hist.float()
        K = hist.shape[1]
        gen = img_n.repeat_interleave(K, dim=0)                                         # cal_scores.py:167
        h = hist.reshape(-1, hist.shape[-1])
        h = h / h.norm(p=2, dim=-1, keepdim=True)                                       # eval_utils.py:544
        sim = 100 * F.cosine_similarity(gen, h)                                         # :547
        hs = sim.view(-1, K).mean(dim=1)                                                # cal_scores.py:178-179
    return tgt, hs


def clip_vision_forward(p: Policy, sd_in: Dict[str, torch.Tensor], heads: int, pixel_values: torch.Tensor):
    """open_clip VisionTransformer.forward (class-token pooling, no patch dropout, eval mode)."""
    sd = SD(sd_in)
    patch = sd["conv1.weight"].shape[-1]
    x = p.conv(pixel_values, sd["conv1.weight"], None, stride=patch)                    # [B, width, g, g]
    B, W = x.shape[0], x.shape[1]
    x = x.reshape(B, W, -1).permute(0, 2, 1)
    cls = sd["class_embedding"].to(x.dtype) + torch.zeros(B, 1, W, dtype=x.dtype)
    x = torch.cat([cls, x], dim=1) + sd["positional_embedding"].to(x.dtype)
    ln = lambda t, pre: p.layer_norm(t, sd[pre + ".weight"], sd[pre + ".bias"], 1e-5).to(t.dtype)
    x = ln(x, "ln_pre")
    D = W // heads
    i = 0
    while sd.has(f"transformer.resblocks.{i}."):
        pre = f"transformer.resblocks.{i}."
        h = ln(x, pre + "ln_1")
        qkv = p.lin(h, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"])
        q, k, v = qkv.chunk(3, dim=-1)
        o = p.attention(q, k, v, heads, D ** -0.5)
        x = x + p.lin(o, sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"])
        h = ln(x, pre + "ln_2")
        m = p.lin(F.gelu(p.lin(h, sd[pre + "mlp.c_fc.weight"], sd[pre + "mlp.c_fc.bias"])),
                  sd[pre + "mlp.c_proj.weight"], sd[pre + "mlp.c_proj.bias"])
        x = x + m
        i += 1
    pooled = ln(x[:, 0], "ln_post")
    return pooled.to(p.compute) @ sd["proj"].to(p.compute)


def clip_text_forward(p: Policy, sd_in: Dict[str, torch.Tensor], heads: int, text: torch.Tensor):
    """open_clip CLIP.encode_text(text, normalize=False): token + positional embeddings, causal pre-norm Transformer,
    ln_final, the row at the end-of-text token (the highest id: `text.argmax(-1)`), `text_projection`."""
    sd = SD(sd_in)
    c = p.compute
    L = text.shape[1]
    x = sd["token_embedding.weight"].to(c)[text] + sd["positional_embedding"].to(c)[:L]
    ln = lambda t, pre: p.layer_norm(t, sd[pre + ".weight"], sd[pre + ".bias"], 1e-5).to(t.dtype)
    W = x.shape[-1]
    D = W // heads
    i = 0
    while sd.has(f"transformer.resblocks.{i}."):
        pre = f"transformer.resblocks.{i}."
        h = ln(x, pre + "ln_1")
        qkv = p.lin(h, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"])
        q, k, v = qkv.chunk(3, dim=-1)
        o = ops_ref.causal_attention_ref(q.to(c), k.to(c), v.to(c), heads, D ** -0.5, out_dtype=c)
        x = x + p.lin(o, sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"])
        h = ln(x, pre + "ln_2")
        m = p.lin(F.gelu(p.lin(h, sd[pre + "mlp.c_fc.weight"], sd[pre + "mlp.c_fc.bias"])),
                  sd[pre + "mlp.c_proj.weight"], sd[pre + "mlp.c_proj.bias"])
        x = x + m
        i += 1
    x = ln(x, "ln_final")
    pooled = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]
    return pooled.to(c) @ sd["text_projection"].to(c)
