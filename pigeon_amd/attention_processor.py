"""Attention module + processors with the diffusers 0.27.2 AttnProcessor protocol.

The reference switches its UNet to xformers with `unet.enable_xformers_memory_efficient_attention()`
(Pigeon/models/lavit_for_pigeon.py:202); diffusers then routes every `Attention.forward` through
`XFormersAttnProcessor.__call__(attn, hidden_states, encoder_hidden_states, attention_mask, temb)`.
`HipAttnProcessor` is that processor for MI355X: the projections stay hipBLASLt GEMMs, the attention core is the
hand-written MFMA kernel (pigeon_amd.ops.attention), and the `[B, N, heads*D]` projection outputs are consumed in
place — no head_to_batch_dim / batch_to_head_dim copies.
"""
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops

_FUSE_QKV = os.environ.get("PGN_FUSE_QKV", "1") != "0"


class HipAttnProcessor:
    """Drop-in for diffusers' XFormersAttnProcessor / AttnProcessor2_0 on gfx950."""

    def __init__(self, cache_cross_kv: bool = False):
        # cross-attention K/V depend only on encoder_hidden_states, which is constant over the denoising loop
        self.cache_cross_kv = cache_cross_kv
        self._kv_cache = None
        # The reference's unconditional CFG branch conditions on ALL-ZERO embeddings (lavit_for_pigeon.py:710-711).
        # With bias-free to_k / to_v that makes K = V = 0 for those batch rows: uniform softmax over zero values, i.e.
        # the attention output is exactly 0 and the layer's output is exactly to_out's bias.  When the pipeline
        # declares the first `zero_ctx_rows` batch rows to be such rows, cross-attention computes the other rows only
        # (query / key / value projections, attention core, output projection) and writes the bias into the rest:
        # the skipped rows are bit-identical to computing them, the others see the same arithmetic (hipBLASLt may pick
        # another tile for the smaller M), at half the cross-attention work.
        self.zero_ctx_rows = 0

    def reset_cache(self):
        self._kv_cache = None

    @staticmethod
    def _self_qkv(attn: "Attention", hs: torch.Tensor):
        """Self-attention projections as ONE [M, C] x [C, 3C] GEMM over the row-concatenated to_q / to_k / to_v
        weights (cached on the module, rebuilt when a weight is replaced or written): 269 us instead of 3 x 103 us
        at M = 32768, C = 1280 and 349 instead of 440 us at M = 131072, C = 640 (2.5 waves of 256^2 tiles become 7.5).
        Each output element is the same K-long dot product as in the separate GEMMs; q, k, v come back as column
        slices (row pitch 3C), which pgn_attention_fwd takes as they are.  PGN_FUSE_QKV=0 switches it off."""
        q, k, v = attn.to_q, attn.to_k, attn.to_v
        if not (_FUSE_QKV and hs.is_cuda and hs.dtype in (torch.bfloat16, torch.float16)
                and q.bias is None and k.bias is None and v.bias is None
                and q.weight.shape == k.weight.shape == v.weight.shape and q.weight.dtype == hs.dtype
                and not (torch.is_grad_enabled() and (q.weight.requires_grad or k.weight.requires_grad
                                                      or v.weight.requires_grad))):   # the fused copy is detached
            return q(hs), k(hs), v(hs)
        key = tuple((w.data_ptr(), w._version) for w in (q.weight, k.weight, v.weight))
        cached = getattr(attn, "_qkv_fused", None)
        if cached is None or cached[0] != key:
            cached = (key, torch.cat([q.weight.detach(), k.weight.detach(), v.weight.detach()]))
            attn._qkv_fused = cached
        qkv = torch.nn.functional.linear(hs, cached[1])
        c = q.weight.shape[0]
        return qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]

    def __call__(self, attn: "Attention", hidden_states: torch.Tensor,
                 encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, temb: Optional[torch.Tensor] = None,
                 *args, **kwargs) -> torch.Tensor:
        residual = hidden_states
        if getattr(attn, "spatial_norm", None) is not None:
            hidden_states = attn.spatial_norm(hidden_states, temb)
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.reshape(b, c, h * w).transpose(1, 2)
        if getattr(attn, "group_norm", None) is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)

        is_cross = encoder_hidden_states is not None
        k0 = self.zero_ctx_rows
        if (is_cross and 0 < k0 < hidden_states.shape[0] and attention_mask is None and input_ndim == 3
                and attn.to_k.bias is None and attn.to_v.bias is None and not getattr(attn, "norm_cross", None)
                and not attn.residual_connection and attn.rescale_output_factor == 1.0):
            return self._cross_skip_zero_rows(attn, hidden_states, encoder_hidden_states, k0)
        if not is_cross:
            query, key, value = self._self_qkv(attn, hidden_states)
        else:
            query = attn.to_q(hidden_states)
            context = encoder_hidden_states
            if getattr(attn, "norm_cross", None):
                context = attn.norm_encoder_hidden_states(context)
            if self.cache_cross_kv and self._kv_cache is not None and self._kv_cache[0] is encoder_hidden_states:
                key, value = self._kv_cache[1], self._kv_cache[2]
            else:
                key, value = attn.to_k(context), attn.to_v(context)
                if self.cache_cross_kv:
                    self._kv_cache = (encoder_hidden_states, key, value)

        key_bias = None
        if attention_mask is not None:
            # diffusers hands the processor an additive mask broadcastable to [B, heads, Nq, Nk]; the path only
            # ever builds key-padding masks ([B, 1, Nk], lavit_for_pigeon.py:873-874) -> [B, Nk] bias
            m = attention_mask
            if m.dim() == 3 and m.shape[1] == 1:
                m = m[:, 0, :]
            if m.dim() != 2 or m.shape[0] not in (query.shape[0], 1) or m.shape[1] != key.shape[1]:
                raise NotImplementedError("HipAttnProcessor supports key-padding masks [B, Nk] / [B, 1, Nk] only")
            key_bias = m.to(torch.float32).expand(query.shape[0], -1)

        hidden_states = ops.attention(query, key, value, attn.heads, attn.scale, key_bias)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)

        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(b, c, h, w)
            if residual.is_contiguous(memory_format=torch.channels_last) and not residual.is_contiguous():
                hidden_states = hidden_states.contiguous(memory_format=torch.channels_last)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        if attn.rescale_output_factor != 1.0:
            hidden_states = hidden_states / attn.rescale_output_factor
        return hidden_states


    # ---- residual-accumulating form (BasicTransformerBlock) ------------------------------------------------------
    @staticmethod
    def can_accumulate(attn: "Attention", hidden_states: torch.Tensor) -> bool:
        lin = attn.to_out[0]
        return (hidden_states.dim() == 3 and hidden_states.is_cuda
                and hidden_states.dtype in (torch.bfloat16, torch.float16) and isinstance(lin, nn.Linear)
                and lin.out_features == hidden_states.shape[-1] and getattr(attn, "spatial_norm", None) is None
                and getattr(attn, "group_norm", None) is None and not getattr(attn, "norm_cross", None)
                and not attn.residual_connection and attn.rescale_output_factor == 1.0)

    def skipped_rows(self, attn: "Attention", batch: int, encoder_hidden_states: Optional[torch.Tensor],
                     attention_mask: Optional[torch.Tensor]) -> int:
        """How many leading batch rows `accumulate` leaves out of a cross-attention call (the all-zero-context rows of
        the CFG pair): the caller need not compute the LayerNorm that only feeds this call's query for them."""
        k0 = self.zero_ctx_rows
        if (encoder_hidden_states is not None and 0 < k0 < batch and attention_mask is None and attn.to_k.bias is None
                and attn.to_v.bias is None):
            return k0
        return 0

    def accumulate(self, attn: "Attention", hidden_states: torch.Tensor, residual: torch.Tensor,
                   encoder_hidden_states: Optional[torch.Tensor] = None,
                   attention_mask: Optional[torch.Tensor] = None, hidden_row0: int = 0) -> Optional[torch.Tensor]:
        """residual += attention(hidden_states) @ to_out.weight^T IN PLACE (a GEMM with beta = 1: no separate add
        launch) and return to_out's bias, which the caller carries as a pending per-channel vector into the next
        LayerNorm (ops.add_layer_norm col_bias).  Rows of an all-zero context (zero_ctx_rows) receive nothing.
        hidden_row0: `hidden_states` starts at that batch row of `residual` (0, or skipped_rows() when the caller
        normalised only the rows this call reads)."""
        B, N, C = residual.shape
        is_cross = encoder_hidden_states is not None
        k0 = self.skipped_rows(attn, B, encoder_hidden_states, attention_mask)
        if hidden_row0 not in (0, k0) or hidden_states.shape[0] != B - hidden_row0:
            raise ValueError("accumulate: hidden_states must cover the batch rows from 0 or from skipped_rows()")
        hs = hidden_states if hidden_row0 == k0 else hidden_states[k0:]
        if not is_cross:
            query, key, value = self._self_qkv(attn, hs)
        else:
            query = attn.to_q(hs)
            context = encoder_hidden_states[k0:]
            if self.cache_cross_kv and self._kv_cache is not None and self._kv_cache[0] is encoder_hidden_states:
                key, value = self._kv_cache[1], self._kv_cache[2]
            else:
                key, value = attn.to_k(context), attn.to_v(context)
                if self.cache_cross_kv:
                    self._kv_cache = (encoder_hidden_states, key, value)
        key_bias = None
        if attention_mask is not None:
            m = attention_mask
            if m.dim() == 3 and m.shape[1] == 1:
                m = m[:, 0, :]
            if m.dim() != 2 or m.shape[0] not in (B, 1) or m.shape[1] != key.shape[1]:
                raise NotImplementedError("HipAttnProcessor supports key-padding masks [B, Nk] / [B, 1, Nk] only")
            key_bias = m.to(torch.float32).expand(B, -1)[k0:]
        o = ops.attention(query, key, value, attn.heads, attn.scale, key_bias)
        lin = attn.to_out[0]
        residual[k0:].view(-1, C).addmm_(o.view(-1, o.shape[-1]), lin.weight.t())
        return lin.bias

    def _cross_skip_zero_rows(self, attn, hidden_states, encoder_hidden_states, k0):
        B, N, _ = hidden_states.shape
        query = attn.to_q(hidden_states[k0:])
        if self.cache_cross_kv and self._kv_cache is not None and self._kv_cache[0] is encoder_hidden_states:
            key, value = self._kv_cache[1], self._kv_cache[2]
        else:
            key, value = attn.to_k(encoder_hidden_states[k0:]), attn.to_v(encoder_hidden_states[k0:])
            if self.cache_cross_kv:
                self._kv_cache = (encoder_hidden_states, key, value)
        o = ops.attention(query, key, value, attn.heads, attn.scale, None)
        lin = attn.to_out[0]
        out = torch.empty(B, N, lin.out_features, dtype=o.dtype, device=o.device)
        if lin.bias is None:
            out[:k0].zero_()
            torch.mm(o.view(-1, o.shape[-1]), lin.weight.t(), out=out[k0:].view(-1, lin.out_features))
        else:
            out[:k0] = lin.bias
            torch.addmm(lin.bias, o.view(-1, o.shape[-1]), lin.weight.t(), out=out[k0:].view(-1, lin.out_features))
        return out


class WideHeadAttnProcessor:
    """Processor for head sizes beyond the fused kernel's 256: the VAE decoder's single 512-wide head (the reference
    leaves that one on diffusers' default processor: only `self.unet` is switched at lavit_for_pigeon.py:202).  Per
    (sample, head): 16-bit GEMM with fp32 scores (hipBLASLt) -> pgn_softmax_rows -> 16-bit GEMM
    (ops.wide_head_attention: 1.1 ms instead of 3.0 ms per 1024^2 image for torch's fused SDPA), the fused kernel's
    rounding points.  bf16 / fp16 device tensors without a mask only - what the path produces (after `upcast_vae` the
    mid block is back in fp16, lavit_for_pigeon.py:808-813); anything else raises instead of falling back to torch."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *a, **kw):
        if attention_mask is not None:
            raise NotImplementedError("WideHeadAttnProcessor: attention masks are not on the Pigeon path")
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.reshape(b, c, h * w).transpose(1, 2)
        if getattr(attn, "group_norm", None) is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        context = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        B, Nq, C = hidden_states.shape
        H = attn.heads
        q = attn.to_q(hidden_states).view(B, Nq, H, C // H).transpose(1, 2)
        k = attn.to_k(context).view(B, -1, H, C // H).transpose(1, 2)
        v = attn.to_v(context).view(B, -1, H, C // H).transpose(1, 2)
        # ops.wide_head_attention raises on CPU / fp32 tensors
        o = torch.stack([torch.stack([ops.wide_head_attention(q[i, j], k[i, j], v[i, j], attn.scale)
                                      for j in range(H)]) for i in range(B)])
        hidden_states = attn.to_out[1](attn.to_out[0](o.transpose(1, 2).reshape(B, Nq, C)))
        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(b, c, h, w)
            if residual.is_contiguous(memory_format=torch.channels_last) and not residual.is_contiguous():
                hidden_states = hidden_states.contiguous(memory_format=torch.channels_last)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        if attn.rescale_output_factor != 1.0:
            hidden_states = hidden_states / attn.rescale_output_factor
        return hidden_states


class Attention(nn.Module):
    """State-dict compatible with diffusers.models.attention_processor.Attention (to_q/to_k/to_v/to_out.0)."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 bias: bool = False, out_bias: bool = True, norm_num_groups: Optional[int] = None, eps: float = 1e-5,
                 residual_connection: bool = False, rescale_output_factor: float = 1.0, processor=None):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.residual_connection = residual_connection
        self.rescale_output_factor = rescale_output_factor
        self.spatial_norm = None
        self.norm_cross = None
        if norm_num_groups is not None:
            from .layers import GroupNormAct
            self.group_norm = GroupNormAct(norm_num_groups, query_dim, eps=eps, act=False)
        else:
            self.group_norm = None
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(0.0)])
        self.processor = processor if processor is not None else HipAttnProcessor()

    def set_processor(self, processor):
        self.processor = processor

    def get_processor(self):
        return self.processor

    # kept for protocol completeness (third-party processors call them); the HIP processor does not need them
    def head_to_batch_dim(self, tensor: torch.Tensor, out_dim: int = 3) -> torch.Tensor:
        b, n, c = tensor.shape
        tensor = tensor.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3)
        return tensor.reshape(b * self.heads, n, c // self.heads) if out_dim == 3 else tensor

    def batch_to_head_dim(self, tensor: torch.Tensor) -> torch.Tensor:
        bh, n, d = tensor.shape
        tensor = tensor.reshape(bh // self.heads, self.heads, n, d).permute(0, 2, 1, 3)
        return tensor.reshape(bh // self.heads, n, d * self.heads)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim: int = 3):
        if attention_mask is None:
            return None
        if attention_mask.shape[-1] != target_length:
            attention_mask = torch.nn.functional.pad(attention_mask, (0, target_length), value=0.0)
        if out_dim == 3 and attention_mask.shape[0] < batch_size * self.heads:
            attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
        elif out_dim == 4:
            attention_mask = attention_mask.unsqueeze(1).repeat_interleave(self.heads, dim=1)
        return attention_mask

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)
