"""Tensor-level entry points over the C ABI (include/pigeon_hip.h).

Every function takes CUDA(HIP) tensors, enqueues on torch's current stream and returns torch tensors
allocated through torch's caching allocator.  Nothing here computes on the CPU and nothing falls back to
a PyTorch implementation: non-device tensors or unsupported shapes raise.
"""
import os
from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from ._lib import (PGN_BF16, PGN_F16, PGN_F32, PGN_PRED_EPSILON, PGN_PRED_SAMPLE, PGN_PRED_V,
                   PGN_STEP_CLIP_SAMPLE, PGN_STEP_FUSED_FP32, PGN_STEP_USE_CLIPPED_OUTPUT)

_DT = {torch.float32: PGN_F32, torch.bfloat16: PGN_BF16, torch.float16: PGN_F16}
_PRED = {"epsilon": PGN_PRED_EPSILON, "sample": PGN_PRED_SAMPLE, "v_prediction": PGN_PRED_V}

# Codebook ids start after the 32000 text ids + 2 specials (lavit_for_pigeon.py:905)
LAVIT_ID_OFFSET = 32002
LAVIT_MAX_TOKENS = 256  # lavit_for_pigeon.py:907-908


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"pigeon_amd: unsupported dtype {t.dtype}") from None


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pigeon_amd ops run on the MI355X only: got a non-device tensor (no CPU fallback)")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# optional per-launch timing hook used by bench.py: callable(kind:str, work:float, fn) -> result
_timer = None


def set_launch_timer(timer):
    global _timer
    _timer = timer


def _launch(kind, work, fn):
    if _timer is None:
        return fn()
    return _timer(kind, work, fn)


# ------------------------------------------------------------------------------------------------
def token_gather(image_tokens: Union[Sequence[torch.Tensor], torch.Tensor], codebook: torch.Tensor,
                 out_dtype: torch.dtype = torch.bfloat16, id_offset: int = LAVIT_ID_OFFSET,
                 max_tokens: int = LAVIT_MAX_TOKENS, pad_len: Optional[int] = None, packed: bool = True,
                 padded: bool = True, check: bool = False):
    """Fused replacement of generate_image_embeds' id filter + F.embedding + cast (lavit_for_pigeon.py:903-918)
    and of HighresVQDecoder's pad/mask loop (modeling_decoder.py:310-317).

    image_tokens: list of 1-D int64 device tensors (ragged) or one [B, L] int64 tensor.
    Returns dict(token_nums int64 [B], packed [sum_worstcase, dim] (rows past sum(token_nums) are
    unspecified), padded [B, pad_len, dim], mask [B, pad_len]).  No host sync unless check=True.
    """
    lib = _lib.load()
    if isinstance(image_tokens, torch.Tensor):
        if image_tokens.dim() != 2:
            raise ValueError("image_tokens tensor must be [B, L]")
        B, L = image_tokens.shape
        ids = image_tokens.contiguous().view(-1)
        lens = [L] * B
    else:
        B = len(image_tokens)
        lens = [int(t.numel()) for t in image_tokens]
        ids = torch.cat([t.reshape(-1) for t in image_tokens]) if B else torch.empty(0, dtype=torch.int64)
    _dev(ids, codebook)
    if ids.dtype != torch.int64:
        raise TypeError("image_tokens must be int64")
    if codebook.dtype != torch.float32 or codebook.dim() != 2:
        raise TypeError("codebook must be fp32 [num_codes, dim]")
    codebook = codebook.contiguous()
    dev = codebook.device
    num_codes, dim = codebook.shape
    offs, acc = [0], 0
    for n in lens:
        acc += n
        offs.append(acc)
    offsets = torch.tensor(offs, dtype=torch.int32).to(dev, non_blocking=True)
    pad_len = max_tokens if pad_len is None else pad_len
    worst = sum(min(n, max_tokens) for n in lens)
    token_nums = torch.empty(B, dtype=torch.int64, device=dev)
    out_packed = torch.empty(worst, dim, dtype=out_dtype, device=dev) if packed else None
    out_padded = torch.empty(B, pad_len, dim, dtype=out_dtype, device=dev) if padded else None
    out_mask = torch.empty(B, pad_len, dtype=out_dtype, device=dev) if padded else None
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    work = 200.0 * worst  # algorithmic bytes: 8 B id + 128 B row + 64 B bf16 row per token

    def run():
        rc = lib.pgn_token_gather(_ptr(ids), _ptr(offsets), B, _ptr(codebook), num_codes, dim, id_offset, max_tokens,
                                  _DT[out_dtype], _ptr(token_nums), _ptr(out_packed), _ptr(out_padded), _ptr(out_mask),
                                  pad_len, _ptr(status), _stream())
        _lib.check(rc, "pgn_token_gather")

    _launch("gather", work, run)
    if check and int(status.item()) != 0:
        raise IndexError("token_gather: a kept token id is outside the codebook (F.embedding would raise)")
    return {"token_nums": token_nums, "packed": out_packed, "padded": out_padded, "mask": out_mask, "status": status}


def _flatten_tokens(image_tokens):
    if isinstance(image_tokens, torch.Tensor):
        if image_tokens.dim() != 2:
            raise ValueError("image_tokens tensor must be [B, L]")
        B, L = image_tokens.shape
        return image_tokens.contiguous().view(-1), [L] * B
    lens = [int(t.numel()) for t in image_tokens]
    ids = torch.cat([t.reshape(-1) for t in image_tokens]) if lens else torch.empty(0, dtype=torch.int64)
    return ids, lens


def token_embed(image_tokens: Union[Sequence[torch.Tensor], torch.Tensor], codebook: torch.Tensor,
                in_proj_weight: torch.Tensor, in_proj_bias: Optional[torch.Tensor], pos_embed: torch.Tensor,
                id_offset: int = LAVIT_ID_OFFSET, max_tokens: int = LAVIT_MAX_TOKENS, pad_len: Optional[int] = None,
                check: bool = False):
    """token_gather carried on through the token decoder's `in_proj`, padding mask and `pos_embed`
    (modeling_decoder.py:310-319) plus the additive key bias of :100-101, in two launches.

    Returns dict(token_nums int64 [B], x [B, pad_len, E], key_bias fp32 [B, pad_len], status)."""
    lib = _lib.load()
    ids, lens = _flatten_tokens(image_tokens)
    B = len(lens)
    if B == 0:
        raise ValueError("token_embed needs at least one token sequence")
    _dev(ids, codebook, in_proj_weight, pos_embed)
    if ids.dtype != torch.int64:
        raise TypeError("image_tokens must be int64")
    if codebook.dtype != torch.float32 or codebook.dim() != 2:
        raise TypeError("codebook must be fp32 [num_codes, dim]")
    dt = in_proj_weight.dtype
    if dt not in (torch.bfloat16, torch.float16):
        raise TypeError("token_embed computes in bf16 / fp16 (the model dtype)")
    num_codes, dim = codebook.shape
    E = in_proj_weight.shape[0]
    pos = pos_embed.reshape(-1, pos_embed.shape[-1])
    pad_len = max_tokens if pad_len is None else pad_len
    if in_proj_weight.shape != (E, dim) or pos.shape[1] != E or pos.shape[0] < pad_len or pos.dtype != dt:
        raise ValueError("in_proj_weight must be [E, dim] and pos_embed [>= pad_len, E] of the same dtype")
    if in_proj_bias is not None and (in_proj_bias.shape != (E,) or in_proj_bias.dtype != dt):
        raise ValueError("in_proj_bias must be [E] of the weight's dtype")
    codebook, w, pos = codebook.contiguous(), in_proj_weight.contiguous(), pos.contiguous()
    bias = None if in_proj_bias is None else in_proj_bias.contiguous()
    dev = codebook.device
    offs, acc = [0], 0
    for n in lens:
        acc += n
        offs.append(acc)
    offsets = torch.tensor(offs, dtype=torch.int32).to(dev, non_blocking=True)
    token_nums = torch.empty(B, dtype=torch.int64, device=dev)
    x = torch.empty(B, pad_len, E, dtype=dt, device=dev)
    key_bias = torch.empty(B, pad_len, dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    work = float(B * pad_len * E * 2 + pad_len * E * 2 + E * dim * 2)

    def run():
        rc = lib.pgn_token_embed(_ptr(ids), _ptr(offsets), B, _ptr(codebook), num_codes, dim, id_offset, max_tokens,
                                 _DT[dt], _ptr(w), _ptr(bias), _ptr(pos), E, _ptr(token_nums), _ptr(x), _ptr(key_bias),
                                 pad_len, _ptr(status), _stream())
        _lib.check(rc, "pgn_token_embed")

    _launch("token_embed", work, run)
    if check and int(status.item()) != 0:
        raise IndexError("token_embed: a kept token id is outside the codebook (F.embedding would raise)")
    return {"token_nums": token_nums, "x": x, "key_bias": key_bias, "status": status}


# ------------------------------------------------------------------------------------------------
def cfg_ddim_step(noise_pred: Optional[torch.Tensor], sample: torch.Tensor, guidance_scale: float,
                  sqrt_alpha_t: float, sqrt_beta_t: float, sqrt_alpha_prev: float, dir_coef: float,
                  prediction_type: str = "epsilon", eps_uncond: Optional[torch.Tensor] = None,
                  eps_text: Optional[torch.Tensor] = None, clip_sample_range: Optional[float] = None,
                  use_clipped_model_output: bool = False, fused_fp32: bool = False, return_pred_x0: bool = False,
                  out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """chunk(2) + guidance + DDIMScheduler.step (lavit_for_pigeon.py:760-764) in one launch.

    noise_pred: [2B, ...] (uncond first, text second) — or pass eps_uncond / eps_text separately
    (eps_uncond=None means "no guidance").  sample: [B, ...] latents (fp32 in the reference).
    """
    lib = _lib.load()
    if noise_pred is not None:
        half = noise_pred.shape[0] // 2
        if noise_pred.shape[0] != 2 * sample.shape[0]:
            raise ValueError("noise_pred must hold [uncond; text] for every latent")
        noise_pred = noise_pred.contiguous()
        eps_uncond, eps_text = noise_pred[:half], noise_pred[half:]
    _dev(eps_uncond, eps_text, sample)
    eps_text = eps_text.contiguous()
    eps_uncond = None if eps_uncond is None else eps_uncond.contiguous()
    sample = sample.contiguous()
    if eps_text.numel() != sample.numel():
        raise ValueError("eps and sample sizes differ")
    prev = torch.empty_like(sample) if out is None else out
    x0 = torch.empty_like(sample) if return_pred_x0 else None
    flags = (PGN_STEP_FUSED_FP32 if fused_fp32 else 0)
    if clip_sample_range is not None:
        flags |= PGN_STEP_CLIP_SAMPLE
        if use_clipped_model_output:
            flags |= PGN_STEP_USE_CLIPPED_OUTPUT
    n = sample.numel()
    es, ss = eps_text.element_size(), sample.element_size()
    work = float(n * ((2 if eps_uncond is not None else 1) * es + 2 * ss + (ss if x0 is not None else 0)))

    def run():
        rc = lib.pgn_cfg_ddim_step(_ptr(eps_uncond), _ptr(eps_text), _dt(eps_text), _ptr(sample), _ptr(prev), _ptr(x0),
                                   _dt(sample), n, float(guidance_scale), float(sqrt_alpha_t), float(sqrt_beta_t),
                                   float(sqrt_alpha_prev), float(dir_coef), _PRED[prediction_type], flags,
                                   float(clip_sample_range or 0.0), _stream())
        _lib.check(rc, "pgn_cfg_ddim_step")

    _launch("cfg_step", work, run)
    return prev, x0


def cfg_euler_step(noise_pred: Optional[torch.Tensor], sample: torch.Tensor, guidance_scale: float, sigma: float,
                   sigma_next: float, prediction_type: str = "epsilon", eps_uncond: Optional[torch.Tensor] = None,
                   eps_text: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None,
                   fused_fp32: bool = False, return_pred_x0: bool = False):
    """chunk(2) + guidance + EulerDiscreteScheduler.step (gamma = 0) in one launch."""
    lib = _lib.load()
    if noise_pred is not None:
        half = noise_pred.shape[0] // 2
        if noise_pred.shape[0] != 2 * sample.shape[0]:
            raise ValueError("noise_pred must hold [uncond; text] for every latent")
        noise_pred = noise_pred.contiguous()
        eps_uncond, eps_text = noise_pred[:half], noise_pred[half:]
    _dev(eps_uncond, eps_text, sample)
    eps_text = eps_text.contiguous()
    eps_uncond = None if eps_uncond is None else eps_uncond.contiguous()
    sample = sample.contiguous()
    out_dtype = eps_text.dtype if out_dtype is None else out_dtype  # diffusers casts back to model_output.dtype
    prev = torch.empty(sample.shape, dtype=out_dtype, device=sample.device)
    x0 = torch.empty_like(prev) if return_pred_x0 else None
    n = sample.numel()
    work = float(n * ((2 if eps_uncond is not None else 1) * eps_text.element_size() + sample.element_size()
                      + prev.element_size() * (2 if x0 is not None else 1)))

    def run():
        rc = lib.pgn_cfg_euler_step(_ptr(eps_uncond), _ptr(eps_text), _dt(eps_text), _ptr(sample), _dt(sample),
                                    _ptr(prev), _ptr(x0), _DT[out_dtype], n, float(guidance_scale), float(sigma),
                                    float(sigma_next), _PRED[prediction_type], PGN_STEP_FUSED_FP32 if fused_fp32 else 0,
                                    _stream())
        _lib.check(rc, "pgn_cfg_euler_step")

    _launch("cfg_step", work, run)
    return prev, x0


def cfg_expand_scale(latents: torch.Tensor, out_dtype: torch.dtype, divisor: float = 1.0) -> torch.Tensor:
    """torch.cat([latents]*2) + scale_model_input + cast (lavit_for_pigeon.py:747-748) in one launch."""
    lib = _lib.load()
    _dev(latents)
    latents = latents.contiguous()
    out = torch.empty((2 * latents.shape[0],) + tuple(latents.shape[1:]), dtype=out_dtype, device=latents.device)
    n = latents.numel()

    def run():
        rc = lib.pgn_cfg_expand_scale(_ptr(latents), _dt(latents), _ptr(out), _DT[out_dtype], n, float(divisor), _stream())
        _lib.check(rc, "pgn_cfg_expand_scale")

    _launch("cfg_expand", float(n * (latents.element_size() + 2 * out.element_size())), run)
    return out


# ------------------------------------------------------------------------------------------------
def group_norm_silu(x: torch.Tensor, num_groups: int, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                    eps: float = 1e-5, silu: bool = True, out_dtype: Optional[torch.dtype] = None,
                    add_nc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GroupNorm (+SiLU) on an NCHW-contiguous tensor [N, C, *spatial]; fp32 statistics and arithmetic.
    add_nc [N, C]: normalise x + add_nc[:, :, None, None] (rounded to x.dtype first, as the tensor add it replaces)."""
    lib = _lib.load()
    _dev(x, weight, bias, add_nc)
    if x.dim() < 2:
        raise ValueError("group_norm_silu expects [N, C, ...]")
    if (x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16) and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last) and out_dtype in (None, x.dtype)
            and x.shape[1] % 8 == 0 and x.shape[1] // num_groups <= 128):
        return _group_norm_silu_nhwc(x, num_groups, weight, bias, eps, silu, add_nc)
    x = x.contiguous()
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C) if N * C else 0
    if (weight is None) != (bias is None):
        raise ValueError("weight and bias must both be given or both be None")
    adt = PGN_F32
    if weight is not None:
        if weight.dtype != bias.dtype or weight.numel() != C or bias.numel() != C:
            raise ValueError("affine parameters must be [C] and share a dtype")
        weight, bias = weight.contiguous(), bias.contiguous()
        adt = _dt(weight)
    y = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    work = float(x.numel() * (x.element_size() + y.element_size()))

    add_dt = PGN_F32
    if add_nc is not None:
        if add_nc.shape != (N, C):
            raise ValueError("add_nc must be [N, C]")
        add_nc = add_nc.contiguous()
        add_dt = _dt(add_nc)
    ws_bytes = int(lib.pgn_groupnorm_workspace_bytes(N, C, HW, int(num_groups))) if N * C and add_nc is None else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None

    def run():
        rc = lib.pgn_groupnorm_silu_ex(_ptr(x), _dt(x), _ptr(y), _dt(y), _ptr(weight), _ptr(bias), adt, _ptr(add_nc),
                                       add_dt, N, C, HW, int(num_groups), float(eps), 1 if silu else 0, _ptr(ws),
                                       ws_bytes, _stream())
        _lib.check(rc, "pgn_groupnorm_silu_ex")

    _launch("groupnorm_silu" if silu else "groupnorm", work, run)
    return y


def _is_channels_last(x: torch.Tensor) -> bool:
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)    # dense [N, H, W, C] memory


def group_norm_silu_cat(x1: torch.Tensor, x2: torch.Tensor, num_groups: int, weight: Optional[torch.Tensor],
                        bias: Optional[torch.Tensor], eps: float = 1e-5, silu: bool = True) -> torch.Tensor:
    """group_norm_silu(torch.cat([x1, x2], dim=1)) for two channels_last activations, without writing the concatenation
    (pgn_groupnorm_silu_nhwc_cat; bit-identical to the two-step form).  Returns a channels_last tensor."""
    _dev(x1, x2, weight, bias)
    if (x1.dim() != 4 or x2.dim() != 4 or x1.shape[0] != x2.shape[0] or x1.shape[2:] != x2.shape[2:]
            or x1.dtype != x2.dtype):
        raise ValueError("group_norm_silu_cat expects two [N, C_i, H, W] tensors of one dtype and spatial shape")
    if x1.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError("group_norm_silu_cat: bf16 / fp16 only")
    if not (_is_channels_last(x1) and _is_channels_last(x2)):
        raise ValueError("group_norm_silu_cat expects channels_last tensors")
    return _group_norm_silu_nhwc(x1, num_groups, weight, bias, eps, silu, None, x2=x2)


def _gn_nhwc_act(silu: bool) -> int:
    """`act` of the channels-last GroupNorm entries: bit 0 SiLU + the choice of form (include/pigeon_hip.h,
    PGN_GN_ONEPASS_*).  The library reads no environment; the two switches tests and measurement scripts flip inside a
    process are read HERE, per call: PGN_GN_NHWC_ONEPASS=0 never / =1 whenever the shape allows,
    PGN_GN_NHWC_ONEPASS_TIMEOUT=0 zero wait bound (every workgroup takes the recompute safety net)."""
    act = 1 if silu else 0
    sw = os.environ.get("PGN_GN_NHWC_ONEPASS", "")
    if sw[:1] == "0":
        act |= 2
    elif sw[:1] == "1":
        act |= 4
    if os.environ.get("PGN_GN_NHWC_ONEPASS_TIMEOUT", "").strip() == "0":
        act |= 8
    pp = os.environ.get("PGN_GN_NHWC_PIPELINE", "")      # 0: three launches for large tensors too, 1: pipelined form always
    if pp[:1] == "0":
        act |= 16
    elif pp[:1] == "1":
        act |= 32
    return act


# Producer-side GroupNorm statistics are emitted for activations ABOVE this size: up to 32 MiB the consuming GroupNorm
# runs as ONE launch (one-pass form, 1 read + 1 write already) and partials would only add a finalize launch.
GN_STATS_MIN_BYTES = 32 << 20


def _gn_stats_bytes(N: int, C: int, HW: int, elem: int) -> int:
    if N * C * HW * elem <= GN_STATS_MIN_BYTES:
        return 0
    return int(_lib.load().pgn_bias_residual_add_stats_bytes(N, C, HW))


def gn_stats_applicable(x: torch.Tensor) -> bool:
    """Would bias_residual_add(x, ..., gn_stats=True) emit partial statistics for this activation?"""
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and _is_channels_last(x) and not x.is_contiguous()):
        return False
    N, C, H, W = x.shape
    return _gn_stats_bytes(N, C, H * W, x.element_size()) > 0


def gn_partials(t: Optional[torch.Tensor]):
    """The partial GroupNorm statistics that rode along with `t` from the launch that produced it
    (bias_residual_add(..., gn_stats=True)), or None: none were emitted, or `t` has been written since."""
    hit = getattr(t, "_pgn_gn_part", None) if t is not None else None
    if hit is None or hit[0] != t._version or hit[1] != t.data_ptr():
        return None
    return hit[2]


def _group_norm_silu_nhwc(x, num_groups, weight, bias, eps, silu, add_nc, x2=None):
    """channels_last input -> channels_last output through pgn_groupnorm_silu_nhwc (no layout change, no copy).
    Inputs that bring their producer's partial statistics (gn_partials) skip the statistics launch
    (pgn_groupnorm_silu_nhwc_pre: one read + one write of the tensor instead of two + one)."""
    lib = _lib.load()
    N, C1, H, W = x.shape
    C = C1 + (x2.shape[1] if x2 is not None else 0)
    HW = H * W
    pre = None
    if add_nc is None:
        p1, p2 = gn_partials(x), gn_partials(x2)
        if p1 is not None and (x2 is None or p2 is not None):
            if (p1.numel() == int(lib.pgn_bias_residual_add_stats_bytes(N, C1, HW)) and
                    (x2 is None or p2.numel() == int(lib.pgn_bias_residual_add_stats_bytes(N, C - C1, HW)))):
                pre = (p1, p2)
    if (weight is None) != (bias is None):
        raise ValueError("weight and bias must both be given or both be None")
    adt = PGN_F32
    if weight is not None:
        if weight.dtype != bias.dtype or weight.numel() != C or bias.numel() != C:
            raise ValueError("affine parameters must be [C] and share a dtype")
        weight, bias = weight.contiguous(), bias.contiguous()
        adt = _dt(weight)
    add_dt = PGN_F32
    if add_nc is not None:
        if add_nc.shape != (N, C):
            raise ValueError("add_nc must be [N, C]")
        add_nc = add_nc.contiguous()
        add_dt = _dt(add_nc)
    if x2 is None:
        y = torch.empty_like(x)     # preserves the channels_last strides
    else:
        y = torch.empty((N, C, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ws_bytes = 8 * N * int(num_groups) if pre else int(lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, int(num_groups)))
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=x.device)
    act = _gn_nhwc_act(silu)

    def run():
        if pre is not None:
            rc = lib.pgn_groupnorm_silu_nhwc_pre(_ptr(x), C1, _ptr(pre[0]), _ptr(x2), C - C1, _ptr(pre[1]), _ptr(y),
                                                 _dt(x), _ptr(weight), _ptr(bias), adt, N, HW, int(num_groups),
                                                 float(eps), 1 if silu else 0, _ptr(ws), ws_bytes, _stream())
        elif x2 is None:
            rc = lib.pgn_groupnorm_silu_nhwc(_ptr(x), _ptr(y), _dt(x), _ptr(weight), _ptr(bias), adt, _ptr(add_nc),
                                             add_dt, N, C, HW, int(num_groups), float(eps), act, _ptr(ws),
                                             ws_bytes, _stream())
        else:
            rc = lib.pgn_groupnorm_silu_nhwc_cat(_ptr(x), C1, _ptr(x2), C - C1, _ptr(y), _dt(x), _ptr(weight),
                                                 _ptr(bias), adt, _ptr(add_nc), add_dt, N, HW, int(num_groups),
                                                 float(eps), act, _ptr(ws), ws_bytes, _stream())
        _lib.check(rc, "pgn_groupnorm_silu_nhwc")

    _launch("groupnorm_silu" if silu else "groupnorm", float(2 * y.numel() * y.element_size()), run)
    return y


def upsample_nearest2x(x: torch.Tensor) -> torch.Tensor:
    """F.interpolate(x, scale_factor=2.0, mode="nearest") for a channels_last [N, C, H, W] activation (bf16 / fp16,
    C % 8 == 0), through pgn_upsample_nearest2x_nhwc; returns a channels_last tensor."""
    lib = _lib.load()
    _dev(x)
    if not _is_channels_last(x):
        raise ValueError("upsample_nearest2x expects a channels_last [N, C, H, W] tensor")
    if x.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError("upsample_nearest2x: bf16 / fp16 only")
    N, C, H, W = x.shape
    y = torch.empty((N, C, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)

    def run():
        rc = lib.pgn_upsample_nearest2x_nhwc(_ptr(x), _ptr(y), _dt(x), N, H, W, C, _stream())
        _lib.check(rc, "pgn_upsample_nearest2x_nhwc")

    _launch("upsample", float(5 * x.numel() * x.element_size()), run)
    return y


# ------------------------------------------------------------------------------------------------
def geglu(h: torch.Tensor) -> torch.Tensor:
    """hidden * gelu(gate) with [hidden | gate] = h.chunk(2, dim=-1) (diffusers GEGLU.forward) in one launch."""
    lib = _lib.load()
    _dev(h)
    if h.shape[-1] % 2:
        raise ValueError("geglu expects an even last dimension")
    D = h.shape[-1] // 2
    h2 = h.reshape(-1, 2 * D)
    if h2.stride(1) != 1:
        h2 = h2.contiguous()
    M = h2.shape[0]
    y = torch.empty(h.shape[:-1] + (D,), dtype=h.dtype, device=h.device)

    def run():
        rc = lib.pgn_geglu(_ptr(h2), _ptr(y), _dt(h), M, D, h2.stride(0), D, _stream())
        _lib.check(rc, "pgn_geglu")

    _launch("geglu", float(3 * M * D * h.element_size()), run)
    return y


def bias_residual_add(x: torch.Tensor, bias: torch.Tensor, residual: Optional[torch.Tensor] = None,
                      inplace: bool = True, gn_stats: bool = False) -> torch.Tensor:
    """x + bias[c] (+ residual) for a [N, C, H, W] activation in NCHW or channels_last memory, one launch.
    `x` is overwritten unless inplace=False (it is the fresh output of a bias-less convolution).

    gn_stats: the result feeds a channels-last GroupNorm.  Where the library's form applies (channels_last, tensor above
    32 MiB: pgn_bias_residual_add_stats) the same launch emits that GroupNorm's partial statistics; they ride on the
    returned tensor (a NEW tensor then: the launch does not work in place) and group_norm_silu / group_norm_silu_cat
    pick them up as long as the tensor has not been written since (gn_partials).  Values are bit-identical either way."""
    lib = _lib.load()
    _dev(x, bias, residual)
    if x.dim() != 4:
        raise ValueError("bias_residual_add expects [N, C, H, W]")
    N, C, H, W = x.shape
    if x.is_contiguous():
        rows, inner = N, H * W
    elif x.is_contiguous(memory_format=torch.channels_last):
        rows, inner = N * H * W, 1
    else:
        raise ValueError("bias_residual_add: x must be NCHW- or channels_last-contiguous")
    if residual is not None:
        if residual.shape != x.shape or residual.dtype != x.dtype:
            raise ValueError("residual must match x")
        if residual.stride() != x.stride():
            residual = residual.contiguous(memory_format=torch.channels_last if inner == 1 else torch.contiguous_format)
    if bias.numel() != C:
        raise ValueError("bias must be [C]")
    bias = bias.contiguous()
    work = float(x.numel() * x.element_size() * (3 if residual is not None else 2))
    if gn_stats and inner == 1 and x.dtype in (torch.bfloat16, torch.float16) and bias.data_ptr() % 16 == 0:
        part_bytes = _gn_stats_bytes(N, C, H * W, x.element_size())
        if part_bytes:
            y = torch.empty_like(x)
            part = torch.empty(part_bytes, dtype=torch.uint8, device=x.device)

            def run_stats():
                rc = lib.pgn_bias_residual_add_stats(_ptr(x), _ptr(residual), _ptr(bias), _dt(bias), _ptr(y), _dt(x), N, C,
                                                     H * W, _ptr(part), part_bytes, _stream())
                _lib.check(rc, "pgn_bias_residual_add_stats")

            _launch("bias_residual", work, run_stats)
            y._pgn_gn_part = (y._version, y.data_ptr(), part)
            return y
    y = x if inplace else torch.empty_like(x)
    if inplace and getattr(x, "_pgn_gn_part", None) is not None:
        del x._pgn_gn_part          # written through a raw pointer: torch's version counter does not see it

    def run():
        rc = lib.pgn_bias_residual_add(_ptr(x), _ptr(residual), _ptr(bias), _dt(bias), _ptr(y), _dt(x), rows, C, inner,
                                       _stream())
        _lib.check(rc, "pgn_bias_residual_add")

    _launch("bias_residual", work, run)
    return y


def clip_scores(image_features: torch.Tensor, text_features: Optional[torch.Tensor] = None,
                history_features: Optional[torch.Tensor] = None):
    """(tgt_score [M], hist_score [M]) fp32: 100*cos(image, text) and the mean over the history images of
    100*cos(image, history) (Evaluation/eval_utils.py:121-129, 540-548; cal_scores.py:178-179).  Either may be None."""
    lib = _lib.load()
    _dev(image_features, text_features, history_features)
    if image_features.dim() != 2:
        raise ValueError("image_features must be [M, D]")
    M, D = image_features.shape
    img = image_features if image_features.stride(1) == 1 else image_features.contiguous()
    txt = None
    if text_features is not None:
        if text_features.shape != (M, D):
            raise ValueError("text_features must be [M, D]")
        txt = text_features if text_features.stride(1) == 1 else text_features.contiguous()
    hist, n_hist = None, 0
    if history_features is not None:
        if history_features.dim() != 3 or history_features.shape[0] != M or history_features.shape[2] != D:
            raise ValueError("history_features must be [M, n_hist, D]")
        hist = history_features.to(torch.float32).contiguous()
        n_hist = hist.shape[1]
    if txt is None and n_hist == 0:
        raise ValueError("clip_scores needs text_features and / or history_features")
    tgt = torch.empty(M, dtype=torch.float32, device=img.device) if txt is not None else None
    hs = torch.empty(M, dtype=torch.float32, device=img.device) if n_hist else None
    work = float(M * D * (img.element_size() + (txt.element_size() if txt is not None else 0) + 4 * n_hist))

    def run():
        rc = lib.pgn_clip_scores(_ptr(img), _dt(img), img.stride(0), _ptr(txt), _dt(txt) if txt is not None else 0,
                                 txt.stride(0) if txt is not None else 0, _ptr(hist), n_hist * D, n_hist, _ptr(tgt),
                                 _ptr(hs), M, D, _stream())
        _lib.check(rc, "pgn_clip_scores")

    _launch("clip_scores", work, run)
    return tgt, hs


def softmax_rows(scores: torch.Tensor, scale: float, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """softmax(scale * scores) over the last dim of an fp32 [M, N] matrix, written in `out_dtype` (bf16 / fp16)."""
    lib = _lib.load()
    _dev(scores)
    if scores.dim() != 2 or scores.dtype != torch.float32:
        raise TypeError("softmax_rows expects an fp32 [M, N] matrix")
    M, N = scores.shape
    if scores.stride(1) != 1 or scores.stride(0) % 4 != 0 or scores.data_ptr() % 16 != 0:
        # rows must start on 16 bytes: re-pitch into a buffer whose leading dimension is a multiple of 4
        pitched = torch.empty(M, (N + 3) // 4 * 4, dtype=torch.float32, device=scores.device)[:, :N]
        pitched.copy_(scores)
        scores = pitched
    out = torch.empty(M, (N + 3) // 4 * 4, dtype=out_dtype, device=scores.device)[:, :N]

    def run():
        rc = lib.pgn_softmax_rows(_ptr(scores), scores.stride(0), _ptr(out), _DT[out_dtype], out.stride(0), M, N,
                                  float(scale), _stream())
        _lib.check(rc, "pgn_softmax_rows")

    _launch("softmax_rows", float(M * N * (4 + out.element_size())), run)
    return out


def wide_head_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float,
                        rows_per_chunk: int = 8192) -> torch.Tensor:
    """softmax(q k^T * scale) v for ONE head whose dim exceeds the fused kernel's 256 (the VAE's 512): per chunk of
    query rows a 16-bit GEMM with fp32 output (hipBLASLt), pgn_softmax_rows, a second GEMM.  q: [Nq, D], k/v: [Nk, D],
    bf16 / fp16.  The score chunk (rows_per_chunk x Nk fp32) is the only intermediate: 512 MiB at Nk = 16384."""
    _dev(q, k, v)
    if q.dim() != 2 or k.dim() != 2 or v.dim() != 2 or q.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError("wide_head_attention expects 16-bit [N, D] matrices")
    out = torch.empty(q.shape[0], v.shape[1], dtype=q.dtype, device=q.device)
    kt = k.t()
    for r0 in range(0, q.shape[0], rows_per_chunk):
        s = torch.mm(q[r0:r0 + rows_per_chunk], kt, out_dtype=torch.float32)
        p = softmax_rows(s, scale, q.dtype)
        del s
        torch.mm(p, v, out=out[r0:r0 + rows_per_chunk])
    return out


def add_layer_norm(x: torch.Tensor, residual: Optional[torch.Tensor], weight: Optional[torch.Tensor],
                   bias: Optional[torch.Tensor], eps: float = 1e-5, col_bias: Optional[torch.Tensor] = None):
    """LayerNorm over the last dim of (x + col_bias + residual).  Returns (sum, normed); sum is x itself when residual
    is None (a pending `col_bias` [C] then stays pending: it is applied to the norm's input only)."""
    lib = _lib.load()
    _dev(x, residual, weight, bias)
    C = x.shape[-1]
    x = x.contiguous()
    M = x.numel() // C
    if residual is not None:
        if residual.shape != x.shape or residual.dtype != x.dtype:
            raise ValueError("residual must match x")
        residual = residual.contiguous()
    if (weight is None) != (bias is None):
        raise ValueError("weight and bias must both be given or both be None")
    adt = PGN_F32
    if weight is not None:
        if weight.numel() != C or bias.numel() != C or weight.dtype != bias.dtype:
            raise ValueError("affine parameters must be [C] and share a dtype")
        weight, bias = weight.contiguous(), bias.contiguous()
        adt = _dt(weight)
    y = torch.empty_like(x)
    s = torch.empty_like(x) if residual is not None else None
    if col_bias is not None:
        if col_bias.numel() != C:
            raise ValueError("col_bias must be [C]")
        col_bias = col_bias.to(torch.float32).contiguous()

    def run():
        rc = lib.pgn_add_layernorm_cb(_ptr(x), _ptr(residual), _ptr(s), _ptr(y), _ptr(weight), _ptr(bias), _dt(x), adt,
                                      _ptr(col_bias), M, C, float(eps), _stream())
        _lib.check(rc, "pgn_add_layernorm_cb")

    _launch("layernorm", float(x.numel() * x.element_size() * (4 if residual is not None else 2)), run)
    return (s if residual is not None else x), y


def image_to_uint8(x: torch.Tensor) -> torch.Tensor:
    """[N, C, H, W] decoder output in [-1, 1] -> [N, H, W, C] uint8, the reference's post-processing
    (lavit_for_pigeon.py:780-782,929) in one launch, byte-identical."""
    lib = _lib.load()
    _dev(x)
    if x.dim() != 4:
        raise ValueError("image_to_uint8 expects [N, C, H, W]")
    x = x.contiguous()
    N, C, H, W = x.shape
    y = torch.empty(N, H, W, C, dtype=torch.uint8, device=x.device)

    def run():
        rc = lib.pgn_image_to_uint8(_ptr(x), _dt(x), _ptr(y), N, C, H, W, _stream())
        _lib.check(rc, "pgn_image_to_uint8")

    _launch("image_u8", float(x.numel() * (x.element_size() + 1)), run)
    return y


# ------------------------------------------------------------------------------------------------
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None,
              key_bias: Optional[torch.Tensor] = None, causal: bool = False) -> torch.Tensor:
    """softmax(q k^T * scale + key_bias) v per head, replacing xformers.ops.memory_efficient_attention.

    q: [B, Nq, heads*D] (the un-permuted output of to_q), k/v: [B, Nk, heads*D]; any strides as long as the
    last dim is contiguous.  key_bias: fp32 [B, Nk] additive (key padding) or None.  Returns [B, Nq, heads*D].
    causal=True (self-attention of a short sequence, the CLIP text tower): key j contributes to query i only for
    j <= i; runs pgn_attention_causal_fwd (N <= 128, no key_bias).
    """
    lib = _lib.load()
    _dev(q, k, v, key_bias)
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3:
        raise ValueError("attention expects [B, N, heads*D] tensors")
    B, Nq, C = q.shape
    Nk = k.shape[1]
    if C % heads or k.shape[2] != C or v.shape[2] != C or v.shape[1] != Nk or k.shape[0] != B or v.shape[0] != B:
        raise ValueError("attention: inconsistent q/k/v shapes")
    D = C // heads
    if not (q.dtype == k.dtype == v.dtype):
        raise TypeError("attention: q, k, v must share a dtype")

    def fix(t):
        return t if t.stride(2) == 1 else t.contiguous()

    q, k, v = fix(q), fix(k), fix(v)
    out = torch.empty(B, Nq, C, dtype=q.dtype, device=q.device)
    if key_bias is not None:
        if key_bias.shape != (B, Nk):
            raise ValueError("key_bias must be [B, Nk]")
        key_bias = key_bias.to(torch.float32).contiguous()
    scale = float(D) ** -0.5 if scale is None else float(scale)
    work = 4.0 * B * heads * Nq * Nk * D
    if causal:
        if key_bias is not None or Nq != Nk:
            raise ValueError("causal attention is self-attention without a key bias")

        def run_causal():
            rc = lib.pgn_attention_causal_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _dt(q), B, heads, Nq, D,
                                              q.stride(0), q.stride(1), D, k.stride(0), k.stride(1), D,
                                              v.stride(0), v.stride(1), D, out.stride(0), out.stride(1), D,
                                              scale, _stream())
            _lib.check(rc, "pgn_attention_causal_fwd")

        _launch("attention_causal", work / 2, run_causal)
        return out

    def run():
        rc = lib.pgn_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _dt(q), B, heads, Nq, Nk, D,
                                   q.stride(0), q.stride(1), D, k.stride(0), k.stride(1), D,
                                   v.stride(0), v.stride(1), D, out.stride(0), out.stride(1), D,
                                   _ptr(key_bias), Nk if key_bias is not None else 0, scale, _stream())
        _lib.check(rc, "pgn_attention_fwd")

    # timing class = the kernel the library picks (pgn_attention_fwd's heuristic, attention.hip): head dim 64 without a
    # bias and with >= 16 key tiles goes to the 64-rows-per-wave LDS-DMA kernel
    if D != 64:
        kind = f"attention_d{D}"
    elif lib.pgn_attention_kernel_class(B, heads, Nq, Nk, D, 0 if key_bias is None else 1) == 1:
        kind = "attention_w64"      # the library's own rule (PGN_ATT_KERNEL_RING64): the 64-rows-per-wave ring kernel
    else:
        kind = "attention"
    _launch(kind, work, run)
    return out
