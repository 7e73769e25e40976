"""Op-level oracle (plain torch on CPU, fp32 unless the reference itself rounds).  TEST INFRASTRUCTURE."""
import math
from typing import List, Optional

import torch
import torch.nn.functional as F


# ---- token gather ------------------------------------------------------------------------------
def token_gather_ref(image_tokens: List[torch.Tensor], codebook: torch.Tensor, dtype=torch.bfloat16,
                     id_offset: int = 32002, max_tokens: int = 256):
    """Pigeon/models/lavit_for_pigeon.py:903-918 — per-sample filter, truncate, concat, F.embedding, cast."""
    pruned, nums = [], []
    for tok in image_tokens:
        tok = tok - id_offset                      # :905
        tok = tok[tok >= 0]                        # :906
        if len(tok) > max_tokens:                  # :907-908
            tok = tok[:max_tokens]
        nums.append(len(tok))
        pruned.append(tok)
    flat = torch.cat(pruned, dim=0) if pruned else torch.empty(0, dtype=torch.long)
    token_nums = torch.as_tensor(nums, dtype=torch.long)
    emb = F.embedding(flat, codebook)              # :916 (CodebookEmbedding.forward, modeling_visual_tokenzier.py:63-64)
    return emb.to(dtype), token_nums               # :918


def pad_and_mask_ref(x: torch.Tensor, token_nums: torch.Tensor, pad_len: Optional[int] = None):
    """Pigeon/models/modeling_decoder.py:310-317 — split, right-pad with zeros, 0/1 mask in x.dtype."""
    B = len(token_nums)
    parts = torch.split(x, token_nums.tolist(), dim=0)
    width = int(token_nums.max().item()) if pad_len is None else pad_len
    x_pad = torch.zeros(B, width, x.shape[1], dtype=x.dtype)
    mask = torch.zeros(B, width, dtype=x.dtype)
    for i, part in enumerate(parts):
        x_pad[i][: len(part)] = part
        mask[i][: len(part)] = 1
    return x_pad, mask


# ---- CFG combine + scheduler steps -------------------------------------------------------------
def cfg_combine_ref(noise_pred: torch.Tensor, guidance_scale: float) -> torch.Tensor:
    """lavit_for_pigeon.py:760-761, evaluated in noise_pred's own dtype (bf16 under autocast): each of the three
    tensor ops rounds to that dtype.  The python scalar enters the GPU kernel as an fp32 `opmath` value (see _cmul);
    torch's CPU backend agrees for bf16 but rounds it to fp16 first for half tensors, hence the explicit form."""
    uncond, text = noise_pred.chunk(2)
    if noise_pred.dtype in (torch.bfloat16, torch.float16):
        g = torch.tensor(guidance_scale, dtype=torch.float32)
        diff = text - uncond
        scaled = (g * diff.to(torch.float32)).to(noise_pred.dtype)
        return uncond + scaled
    return uncond + guidance_scale * (text - uncond)


def _cmul(coef: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """`0-dim fp32 tensor * tensor` AS THE REFERENCE'S PLATFORM EVALUATES IT.  The result dtype is t.dtype
    (a 0-dim tensor does not promote a dimensioned one).  On CUDA/HIP the coefficient is kept in fp32
    (TensorIterator hands CPU scalars to the kernel as opmath_t) and only the product is rounded to t.dtype;
    torch's CPU backend instead rounds the COEFFICIENT to bf16 first.  The reference runs this line on the GPU
    (lavit_for_pigeon.py:764, scheduler.step on CUDA tensors), so the oracle spells the GPU semantics out."""
    if t.dtype in (torch.bfloat16, torch.float16):
        return (coef.to(torch.float32) * t.to(torch.float32)).to(t.dtype)
    return coef * t


def ddim_step_ref(model_output, sample, alpha_prod_t, alpha_prod_t_prev, prediction_type="epsilon",
                  clip_sample_range=None, use_clipped_model_output=False):
    """diffusers 0.27.2 DDIMScheduler.step with eta = 0, written with the same tensor expressions so torch's
    type promotion puts the roundings where the reference has them (0-dim fp32 coefficient * bf16 tensor
    stays bf16; fp32 sample - bf16 tensor is fp32).  alpha_prod_* are 0-dim fp32 tensors.  PARITY UNPINNED."""
    beta_prod_t = 1 - alpha_prod_t
    if prediction_type == "epsilon":
        x0 = (sample - _cmul(beta_prod_t ** 0.5, model_output)) / alpha_prod_t ** 0.5
        eps = model_output
    elif prediction_type == "sample":
        x0 = model_output
        eps = (sample - _cmul(alpha_prod_t ** 0.5, x0)) / beta_prod_t ** 0.5
    elif prediction_type == "v_prediction":
        x0 = _cmul(alpha_prod_t ** 0.5, sample) - _cmul(beta_prod_t ** 0.5, model_output)
        eps = _cmul(alpha_prod_t ** 0.5, model_output) + _cmul(beta_prod_t ** 0.5, sample)
    else:
        raise ValueError(prediction_type)
    if clip_sample_range is not None:
        x0 = x0.clamp(-clip_sample_range, clip_sample_range)
        if use_clipped_model_output:
            eps = (sample - _cmul(alpha_prod_t ** 0.5, x0)) / beta_prod_t ** 0.5
    std_dev_t = 0.0
    direction = _cmul((1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5, eps)
    prev = _cmul(alpha_prod_t_prev ** 0.5, x0) + direction
    return prev, x0


def euler_step_ref(model_output, sample, sigma, sigma_next, prediction_type="epsilon"):
    """diffusers 0.27.2 EulerDiscreteScheduler.step with s_churn = 0 (gamma = 0); sigma* are 0-dim fp32 tensors.
    PARITY UNPINNED (third-party, restated from the release's published behaviour)."""
    sample = sample.to(torch.float32)
    sigma_hat = sigma
    if prediction_type == "epsilon":
        x0 = sample - _cmul(sigma_hat, model_output)
    elif prediction_type in ("sample", "original_sample"):
        x0 = model_output
    elif prediction_type == "v_prediction":
        x0 = _cmul(-sigma / (sigma ** 2 + 1) ** 0.5, model_output) + (sample / (sigma ** 2 + 1))
    else:
        raise ValueError(prediction_type)
    derivative = (sample - x0) / sigma_hat
    dt = sigma_next - sigma_hat
    prev = sample + derivative * dt
    return prev.to(model_output.dtype), x0


def ddim_step_explicit(eps_u, eps_t, sample, g, sa_t, sb_t, sa_prev, dirc, edt):
    """The kernel's documented arithmetic (epsilon prediction) written with explicit roundings; used by the
    CPU suite to show that it equals torch's own mixed-dtype evaluation above."""
    def r(x):
        return x.to(edt).to(torch.float32)
    eu, et, x = eps_u.float(), eps_t.float(), sample.float()
    e = r(eu + r(g * r(et - eu)))
    x0 = (x - r(sb_t * e)) / sa_t
    return sa_prev * x0 + r(dirc * e), x0


# ---- GroupNorm + SiLU ----------------------------------------------------------------------------
def group_norm_silu_ref(x, num_groups, weight, bias, eps=1e-5, silu=True, out_dtype=None):
    """What autocast makes of diffusers' `nonlinearity(norm(x))`: group_norm in fp32, SiLU in fp32, then the
    cast the following conv/linear applies.  F.group_norm / F.silu are the very ops diffusers calls."""
    y = F.group_norm(x.float(), num_groups, None if weight is None else weight.float(),
                     None if bias is None else bias.float(), eps)
    if silu:
        y = F.silu(y)
    return y.to(out_dtype or x.dtype)


# ---- attention -----------------------------------------------------------------------------------
def attention_ref(q, k, v, heads, scale=None, key_bias=None, out_dtype=None):
    """xformers.ops.memory_efficient_attention semantics on [B, N, heads*D] tensors: fp32 softmax/accumulate over
    the (bf16) inputs, output rounded once.  key_bias: [B, Nk] additive."""
    B, Nq, C = q.shape
    Nk = k.shape[1]
    D = C // heads
    scale = D ** -0.5 if scale is None else scale
    qh = q.float().view(B, Nq, heads, D).transpose(1, 2)
    kh = k.float().view(B, Nk, heads, D).transpose(1, 2)
    vh = v.float().view(B, Nk, heads, D).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2)) * scale
    if key_bias is not None:
        s = s + key_bias.float()[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, vh).transpose(1, 2).reshape(B, Nq, C)
    return o.to(out_dtype or q.dtype)


def causal_attention_ref(q, k, v, heads, scale=None, out_dtype=None):
    """nn.MultiheadAttention with open_clip's `attn_mask` (upper triangle -inf, text Transformer): key j reaches query
    i only for j <= i.  fp32 softmax / accumulate over the given inputs, output rounded once."""
    B, N, C = q.shape
    D = C // heads
    scale = D ** -0.5 if scale is None else scale
    qh = q.float().view(B, N, heads, D).transpose(1, 2)
    kh = k.float().view(B, N, heads, D).transpose(1, 2)
    vh = v.float().view(B, N, heads, D).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2)) * scale
    s = s + torch.full((N, N), float("-inf")).triu_(1)
    o = torch.matmul(torch.softmax(s, dim=-1), vh).transpose(1, 2).reshape(B, N, C)
    return o.to(out_dtype or q.dtype)


# ---- transformer-block row kernels -------------------------------------------------------------------------
def geglu_ref(h: torch.Tensor) -> torch.Tensor:
    """diffusers 0.27.2 GEGLU.forward after the projection: `hidden, gate = h.chunk(2, dim=-1); hidden * F.gelu(gate)`,
    evaluated in h's own dtype (bf16 under autocast: gelu rounds, the product rounds).  PARITY UNPINNED."""
    hidden, gate = h.chunk(2, dim=-1)
    return hidden * F.gelu(gate)


def add_layer_norm_ref(x: torch.Tensor, residual: Optional[torch.Tensor], weight, bias, eps: float = 1e-5,
                       col_bias: Optional[torch.Tensor] = None):
    """`s = x + residual` in the io dtype (diffusers BasicTransformerBlock residual adds), then F.layer_norm in fp32
    (autocast runs layer_norm in fp32) rounded once to the io dtype.  Returns (s, y, y_fp32).  PARITY UNPINNED.
    `col_bias` (fp32 [C]): the bias a bias-free accumulating projection still owes — s = x + col_bias (+ residual)
    with one rounding when a residual materialises the sum; without one the norm sees the unrounded fp32 value."""
    if col_bias is None:
        s = x if residual is None else x + residual
        sn = s.float()
    elif residual is None:
        s, sn = x, x.float() + col_bias.float()
    else:
        s = (x.float() + col_bias.float() + residual.float()).to(x.dtype)
        sn = s.float()
    w = None if weight is None else weight.float()
    b = None if bias is None else bias.float()
    y32 = F.layer_norm(sn, (s.shape[-1],), w, b, eps)
    return s, y32.to(x.dtype), y32


def bias_residual_add_ref(x: torch.Tensor, bias: torch.Tensor, residual=None) -> torch.Tensor:
    """What torch does after a bias-less convolution inside diffusers ResnetBlock2D: a broadcast bias add rounded to
    the activation dtype, then `input_tensor + hidden_states` rounded again.  x: [N, C, H, W]."""
    y = (x.float() + bias.float().view(1, -1, 1, 1)).to(x.dtype)
    if residual is not None:
        y = (y.float() + residual.float()).to(x.dtype)
    return y
