"""UNet2DConditionModel with the diffusers 0.27.2 surface the reference uses
(Pigeon/models/lavit_for_pigeon.py:190-191,202,717,752-757,873-874):

    unet(sample, t, encoder_hidden_states=..., added_cond_kwargs={"text_embeds", "time_ids"}).sample
    unet(sample, t, encoder_hidden_states=..., encoder_attention_mask=...).sample        (lowres variant)
    unet.config.in_channels, unet.enable_xformers_memory_efficient_attention(), .requires_grad_(False), .to(device)

Config keys and state-dict names are diffusers', so a LaVIT / SDXL `unet/config.json` +
`diffusion_pytorch_model.{bin,safetensors}` loads unchanged.  Dense work (3x3 convs, Linear / GEGLU) stays on
hipBLASLt / MIOpen through torch; attention cores, GroupNorm(+SiLU) run in the hand-written HIP kernels.  The
module computes in the parameter dtype (bf16) with fp32 statistics inside the norm kernels, which puts the
roundings where `torch.cuda.amp.autocast(bf16)` puts them for the reference (lavit_for_pigeon.py:706).
"""
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import layers, ops
from .attention_processor import Attention, HipAttnProcessor
from .layers import (Downsample2D, GroupNormAct, ResnetBlock2D, TimestepEmbedding, Upsample2D, conv2d_bias,
                     derived_constant,
                     timestep_embedding)


class _Config(dict):
    __getattr__ = dict.__getitem__


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor


SDXL_CONFIG = dict(
    sample_size=128, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    block_out_channels=(320, 640, 1280), layers_per_block=2, downsample_padding=1, act_fn="silu",
    norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=2048, transformer_layers_per_block=(1, 2, 10),
    attention_head_dim=(5, 10, 20), use_linear_projection=True, addition_embed_type="text_time",
    addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
)

# SD-1.5 shape (the reference's `pixel_decoding='lowres'` variant, lavit_for_pigeon.py:822-895)
SD15_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, act_fn="silu",
    norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, transformer_layers_per_block=1,
    attention_head_dim=8, use_linear_projection=False, addition_embed_type=None,
    addition_time_embed_dim=None, projection_class_embeddings_input_dim=None,
)


# A/B switch for measurements: PGN_ACCUMULATE_ATTENTION=0 restores add + LayerNorm after every attention layer
ACCUMULATE_ATTENTION = os.environ.get("PGN_ACCUMULATE_ATTENTION", "1") != "0"
ACCUMULATE_FF = os.environ.get("PGN_ACCUMULATE_FF", "1") != "0"
# PGN_SKIP_UNUSED_NORM2=0: norm2 of a transformer block is computed for every batch row, also for the rows whose
# cross-attention is skipped (all-zero context: the unconditional half of the CFG pair)
SKIP_UNUSED_NORM2 = os.environ.get("PGN_SKIP_UNUSED_NORM2", "1") != "0"


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        # `hidden, gate = proj(x).chunk(2, dim=-1); hidden * F.gelu(gate)` (diffusers GEGLU.forward) in one launch
        return ops.geglu(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_attention_dim: Optional[int]):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def _pending_biases(self, cb_in):
        """Running fp32 [C] totals of the biases owed by the bias-free accumulating GEMMs: what is pending after attn1,
        after attn2 and after the feed-forward, on top of what the previous block left (`cb_in`).  Cached per parameter
        storage, VERSION and incoming vector (inference: frozen weights, the same chain of blocks every call; an
        in-place update - load_state_dict, a LoRA merge - bumps `_version` and rebuilds the totals)."""
        bs = (self.attn1.to_out[0].bias, self.attn2.to_out[0].bias, self.ff.net[2].bias)
        key = tuple(None if t is None else (t.data_ptr(), t.dtype, t._version) for t in bs + (cb_in,))
        if getattr(self, "_cb_key", None) != key:
            tot, out = cb_in, []
            for t in bs:
                if t is not None:
                    tot = t.detach().float() if tot is None else tot + t.detach().float()
                out.append(tot)
            self._cb_key, self._cb = key, tuple(out)
        return self._cb

    def forward(self, hidden_states, encoder_hidden_states=None, encoder_attention_mask=None, attention_mask=None,
                pending=None, defer_add=False, col_bias=None, owns_input=False):
        """diffusers BasicTransformerBlock.forward: x += attn1(norm1(x)); x += attn2(norm2(x), ctx); x += ff(norm3(x)).
        `pending` is a tensor still to be added to `hidden_states` (the previous block's feed-forward output) and
        `col_bias` a per-channel fp32 vector still owed to it; with `defer_add` the block returns
        (x, pending, col_bias) and leaves the last adds to its consumer.
        On the HIP processor the attention output projections and the feed-forward's second GEMM ACCUMULATE into the
        residual buffer (GEMM with beta = 1, in place) and their biases ride as a pending fp32 [C] vector into the next
        LayerNorm: every norm reads one tensor and writes one (2 passes) instead of add + norm (4 passes).  Other
        processors keep the add-fused-with-LayerNorm form."""
        n1, n2, n3 = self.norm1, self.norm2, self.norm3
        p1, p2 = self.attn1.processor, self.attn2.processor
        accumulate = (ACCUMULATE_ATTENTION and isinstance(p1, HipAttnProcessor) and isinstance(p2, HipAttnProcessor)
                      and p1.can_accumulate(self.attn1, hidden_states) and p2.can_accumulate(self.attn2, hidden_states))
        if not accumulate and pending is None and col_bias is not None:
            hidden_states = hidden_states + col_bias.to(hidden_states.dtype)      # nothing downstream carries it
            col_bias = None
        hidden_states, h = ops.add_layer_norm(hidden_states, pending, n1.weight, n1.bias, n1.eps, col_bias=col_bias)
        if accumulate:
            if pending is None:
                cb_in = col_bias                          # still pending: norm1 only looked at it
                if not owns_input:
                    hidden_states = hidden_states.clone() # do not accumulate into a tensor the caller may still use
            else:
                cb_in = None                              # materialised into the fresh sum by norm1
            c1, c12, c123 = self._pending_biases(cb_in)
            p1.accumulate(self.attn1, h, hidden_states, None, attention_mask)
            # norm2 feeds attn2's query only, and attn2 leaves out the all-zero-context rows of the CFG pair: their
            # LayerNorm is not computed either (SKIP_UNUSED_NORM2)
            k0 = p2.skipped_rows(self.attn2, hidden_states.shape[0], encoder_hidden_states, encoder_attention_mask) \
                if SKIP_UNUSED_NORM2 else 0
            _, h = ops.add_layer_norm(hidden_states[k0:] if k0 else hidden_states, None, n2.weight, n2.bias, n2.eps,
                                      col_bias=c1)
            p2.accumulate(self.attn2, h, hidden_states, encoder_hidden_states, encoder_attention_mask, hidden_row0=k0)
            _, h = ops.add_layer_norm(hidden_states, None, n3.weight, n3.bias, n3.eps, col_bias=c12)
            lin2 = self.ff.net[2]
            if ACCUMULATE_FF and defer_add and isinstance(lin2, nn.Linear):
                g = self.ff.net[0](h)
                hidden_states.view(-1, hidden_states.shape[-1]).addmm_(g.view(-1, g.shape[-1]), lin2.weight.t())
                return hidden_states, None, c123
            f = self.ff(h)
            if defer_add:
                return hidden_states, f, c12
            out = f + hidden_states
            return out if c12 is None else out + c12.to(out.dtype)
        a = self.attn1(h, attention_mask=attention_mask)
        hidden_states, h = ops.add_layer_norm(a, hidden_states, n2.weight, n2.bias, n2.eps)
        a = self.attn2(h, encoder_hidden_states=encoder_hidden_states, attention_mask=encoder_attention_mask)
        hidden_states, h = ops.add_layer_norm(a, hidden_states, n3.weight, n3.bias, n3.eps)
        f = self.ff(h)
        if defer_add:
            return hidden_states, f, None
        return f + hidden_states


class Transformer2DModel(nn.Module):
    def __init__(self, heads: int, dim_head: int, in_channels: int, num_layers: int, cross_attention_dim: int,
                 norm_num_groups: int, use_linear_projection: bool):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = GroupNormAct(norm_num_groups, in_channels, eps=1e-6, act=False)
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner)
            self.proj_out = nn.Linear(inner, in_channels)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, 1)
            self.proj_out = nn.Conv2d(inner, in_channels, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])

    def forward(self, hidden_states, encoder_hidden_states=None, encoder_attention_mask=None, attention_mask=None):
        b, c, h, w = hidden_states.shape
        residual = hidden_states
        hidden_states = self.norm(hidden_states)
        if self.use_linear_projection:
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, h * w, c)
            hidden_states = self.proj_in(hidden_states)
        else:
            hidden_states = self.proj_in(hidden_states)
            hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, h * w, -1)
        pending, cb = None, None
        for block in self.transformer_blocks:
            hidden_states, pending, cb = block(hidden_states, encoder_hidden_states, encoder_attention_mask,
                                               attention_mask, pending=pending, defer_add=True, col_bias=cb,
                                               owns_input=True)   # proj_in's output is ours
        if pending is not None:
            hidden_states = pending + hidden_states
        if cb is not None and not self.use_linear_projection:
            hidden_states = hidden_states + cb.to(hidden_states.dtype)
        # channels_last activations: [B, HW, C] IS the memory of the [B, C, H, W] tensor, so both reshapes are views;
        # NCHW activations need the transposing copy diffusers makes
        cl = residual.is_contiguous(memory_format=torch.channels_last) and not residual.is_contiguous()
        if self.use_linear_projection:
            po = self.proj_out
            dt = hidden_states.dtype
            b_eff = po.bias
            if cb is not None:
                # the last block's pending bias goes through the projection as a bias term: W (x + cb) + b
                b_eff = derived_constant(
                    self, f"_pgn_b_eff_{dt}", (po.weight, po.bias, cb),
                    lambda: (po.weight.detach().float() @ cb + (0.0 if po.bias is None else po.bias.detach().float())).to(dt))
            if (layers.PRODUCER_STATS and cl and b_eff is not None and b_eff.dtype == dt and po.weight.dtype == dt
                    and ops.gn_stats_applicable(residual)):
                # the module's output feeds the next ResnetBlock2D's norm1 (alone or as one half of an up block's
                # concatenation): bias and residual are added by the launch that also emits that GroupNorm's
                # partial statistics, instead of a GEMM bias epilogue + a torch add (the same three tensor passes)
                hidden_states = torch.nn.functional.linear(hidden_states, po.weight)
                hidden_states = hidden_states.reshape(b, h, w, c).permute(0, 3, 1, 2)
                return ops.bias_residual_add(hidden_states, b_eff, residual, inplace=False, gn_stats=True)
            hidden_states = torch.nn.functional.linear(hidden_states, po.weight, b_eff)
            hidden_states = hidden_states.reshape(b, h, w, c).permute(0, 3, 1, 2)
            if not cl:
                hidden_states = hidden_states.contiguous()
        else:
            hidden_states = hidden_states.reshape(b, h, w, -1).permute(0, 3, 1, 2)
            if not cl:
                hidden_states = hidden_states.contiguous()
            hidden_states = self.proj_out(hidden_states)
        return hidden_states + residual


class _DownBlock(nn.Module):
    def __init__(self, cfg, in_ch, out_ch, temb_ch, add_downsample, cross: bool, heads=None, n_tr=1):
        super().__init__()
        L, G, eps = cfg.layers_per_block, cfg.norm_num_groups, cfg.norm_eps
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, temb_ch, G, eps) for i in range(L)])
        if cross:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, out_ch // heads, out_ch, n_tr, cfg.cross_attention_dim, G,
                                    cfg.use_linear_projection) for _ in range(L)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch, cfg.downsample_padding)]) if add_downsample else None

    def forward(self, hidden_states, temb, encoder_hidden_states=None, encoder_attention_mask=None):
        outputs = ()
        for i, resnet in enumerate(self.resnets):
            hidden_states = resnet(hidden_states, temb)
            if self.attentions is not None:
                hidden_states = self.attentions[i](hidden_states, encoder_hidden_states, encoder_attention_mask)
            outputs += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            outputs += (hidden_states,)
        return hidden_states, outputs


class _MidBlock(nn.Module):
    def __init__(self, cfg, ch, temb_ch, heads, n_tr):
        super().__init__()
        G, eps = cfg.norm_num_groups, cfg.norm_eps
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch, G, eps), ResnetBlock2D(ch, ch, temb_ch, G, eps)])
        self.attentions = nn.ModuleList(
            [Transformer2DModel(heads, ch // heads, ch, n_tr, cfg.cross_attention_dim, G, cfg.use_linear_projection)])

    def forward(self, hidden_states, temb, encoder_hidden_states=None, encoder_attention_mask=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        hidden_states = self.attentions[0](hidden_states, encoder_hidden_states, encoder_attention_mask)
        return self.resnets[1](hidden_states, temb)


class _UpBlock(nn.Module):
    def __init__(self, cfg, in_ch, out_ch, prev_ch, temb_ch, add_upsample, cross: bool, heads=None, n_tr=1):
        super().__init__()
        L, G, eps = cfg.layers_per_block + 1, cfg.norm_num_groups, cfg.norm_eps
        resnets = []
        for i in range(L):
            skip = in_ch if i == L - 1 else out_ch
            rin = prev_ch if i == 0 else out_ch
            resnets.append(ResnetBlock2D(rin + skip, out_ch, temb_ch, G, eps))
        self.resnets = nn.ModuleList(resnets)
        if cross:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, out_ch // heads, out_ch, n_tr, cfg.cross_attention_dim, G,
                                    cfg.use_linear_projection) for _ in range(L)])
        else:
            self.attentions = None
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_upsample else None

    def forward(self, hidden_states, res_samples, temb, encoder_hidden_states=None, encoder_attention_mask=None,
                upsample_size=None):
        for i, resnet in enumerate(self.resnets):
            res = res_samples[-1]
            res_samples = res_samples[:-1]
            hidden_states = resnet(hidden_states, temb, skip=res)
            if self.attentions is not None:
                hidden_states = self.attentions[i](hidden_states, encoder_hidden_states, encoder_attention_mask)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states, upsample_size)
        return hidden_states


def _per_block(value, n):
    return tuple(value) if isinstance(value, (list, tuple)) else (value,) * n


class UNet2DConditionModel(nn.Module):
    def __init__(self, **config):
        super().__init__()
        full = dict(SDXL_CONFIG)
        full.update({k: v for k, v in config.items() if not k.startswith("_")})
        for k in ("class_embed_type", "encoder_hid_dim", "num_attention_heads", "time_embedding_type",
                  "conv_in_kernel", "conv_out_kernel", "resnet_time_scale_shift", "only_cross_attention",
                  "dual_cross_attention", "upcast_attention", "cross_attention_norm", "mid_block_only_cross_attention"):
            full.setdefault(k, None)
        cfg = self.config = _Config(full)
        if cfg.act_fn not in ("silu", "swish"):
            raise NotImplementedError("act_fn other than silu")
        if cfg.class_embed_type is not None or cfg.encoder_hid_dim is not None:
            raise NotImplementedError("class / encoder_hid embeddings are not on the Pigeon path")
        if cfg.resnet_time_scale_shift not in (None, "default"):
            raise NotImplementedError("resnet_time_scale_shift != default")
        chans = tuple(cfg.block_out_channels)
        nb = len(chans)
        heads = _per_block(cfg.num_attention_heads or cfg.attention_head_dim, nb)  # diffusers' historical naming
        n_tr = _per_block(cfg.transformer_layers_per_block, nb)
        temb_ch = chans[0] * 4

        self.conv_in = nn.Conv2d(cfg.in_channels, chans[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(chans[0], temb_ch)
        if cfg.addition_embed_type == "text_time":
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, temb_ch)
        elif cfg.addition_embed_type is not None:
            raise NotImplementedError(f"addition_embed_type {cfg.addition_embed_type}")

        self.down_blocks = nn.ModuleList()
        out_ch = chans[0]
        for i, typ in enumerate(cfg.down_block_types):
            in_ch, out_ch = out_ch, chans[i]
            self.down_blocks.append(_DownBlock(cfg, in_ch, out_ch, temb_ch, i != nb - 1,
                                               typ == "CrossAttnDownBlock2D", heads[i], n_tr[i]))
        if cfg.mid_block_type != "UNetMidBlock2DCrossAttn":
            raise NotImplementedError(cfg.mid_block_type)
        self.mid_block = _MidBlock(cfg, chans[-1], temb_ch, heads[-1], n_tr[-1])

        self.up_blocks = nn.ModuleList()
        rev, rheads, rtr = chans[::-1], heads[::-1], n_tr[::-1]
        out_ch = rev[0]
        for i, typ in enumerate(cfg.up_block_types):
            prev_ch, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, nb - 1)]
            self.up_blocks.append(_UpBlock(cfg, in_ch, out_ch, prev_ch, temb_ch, i != nb - 1,
                                           typ == "CrossAttnUpBlock2D", rheads[i], rtr[i]))
        self.num_upsamplers = nb - 1
        self.conv_norm_out = GroupNormAct(cfg.norm_num_groups, chans[0], eps=cfg.norm_eps, act=True)
        self.conv_out = nn.Conv2d(chans[0], cfg.out_channels, 3, padding=1)

    # ---- diffusers surface ---------------------------------------------------------------------
    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    @property
    def attn_processors(self) -> Dict[str, Any]:
        return {f"{n}.processor": m.processor for n, m in self.named_modules() if isinstance(m, Attention)}

    def set_attn_processor(self, processor):
        for n, m in self.named_modules():
            if isinstance(m, Attention):
                if isinstance(processor, dict):
                    m.set_processor(processor[f"{n}.processor"])
                else:
                    # one processor object per layer: processors may hold per-layer state (cross K/V cache)
                    if isinstance(processor, HipAttnProcessor):
                        new = HipAttnProcessor(processor.cache_cross_kv)
                        new.zero_ctx_rows = processor.zero_ctx_rows
                        m.set_processor(new)
                    else:
                        m.set_processor(processor)

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        """What lavit_for_pigeon.py:202 calls; on MI355X the memory-efficient kernel is the HIP one."""
        self.set_attn_processor(HipAttnProcessor())

    def enable_channels_last(self, enabled: bool = True):
        """Keep activations and convolution weights in torch's channels_last memory format: the layout the MIOpen / CK
        convolution kernels compute in (no NCHW<->NHWC transposes around each convolution), served by the channels-last
        GroupNorm kernel; the [B, HW, C] reshapes of the transformer blocks become views.  Results are unchanged."""
        self.channels_last = enabled
        self.to(memory_format=torch.channels_last if enabled else torch.contiguous_format)
        return self

    def set_zero_context_rows(self, rows: int = 0):
        """Declare that the first `rows` batch rows of every following call carry all-zero `encoder_hidden_states`
        (the reference's unconditional CFG half, lavit_for_pigeon.py:710-711,731): cross-attention then skips them
        — their output is exactly to_out's bias either way (HipAttnProcessor.zero_ctx_rows).  0 switches it off."""
        for m in self.modules():
            if isinstance(m, Attention) and isinstance(m.processor, HipAttnProcessor):
                m.processor.zero_ctx_rows = int(rows)
                m.processor.reset_cache()     # cached cross K/V hold the rows of the previous mode

    def enable_cross_kv_cache(self, enabled: bool = True):
        """Cross-attention K / V are kept per processor and reused while the SAME encoder_hidden_states tensor object
        comes in (a denoising loop); every call of this method drops what is cached."""
        self.cross_kv_cache_enabled = bool(enabled)
        for m in self.modules():
            if isinstance(m, Attention) and isinstance(m.processor, HipAttnProcessor):
                m.processor.cache_cross_kv = enabled
                m.processor.reset_cache()

    @classmethod
    def from_config(cls, config, **overrides):
        cfg = dict(config)
        cfg.update(overrides)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, use_safetensors=None, **unused):
        """Load `config.json` + `diffusion_pytorch_model.{safetensors,bin}` (same files diffusers reads,
        lavit_for_pigeon.py:190-191).  `.bin` files go through torch.load(weights_only=True)."""
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "config.json")) as f:
            cfg = json.load(f)
        model = cls(**cfg)
        st = os.path.join(d, "diffusion_pytorch_model.safetensors")
        pt = os.path.join(d, "diffusion_pytorch_model.bin")
        if os.path.exists(st) and use_safetensors is not False:
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(pt):
            sd = torch.load(pt, map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no diffusion_pytorch_model.safetensors/.bin under {d}")
        model.load_state_dict(sd)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None,
                added_cond_kwargs: Optional[Dict[str, torch.Tensor]] = None,
                encoder_attention_mask: Optional[torch.Tensor] = None, return_dict: bool = True):
        cfg = self.config
        dtype = self.dtype
        factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(s % factor != 0 for s in sample.shape[-2:])

        if encoder_attention_mask is not None:
            # bool/0-1 key mask [B, Nk] -> additive bias [B, 1, Nk] (diffusers UNet2DConditionModel.forward)
            encoder_attention_mask = ((1 - encoder_attention_mask.to(torch.float32)) * -10000.0).unsqueeze(1)
        if attention_mask is not None:
            attention_mask = ((1 - attention_mask.to(torch.float32)) * -10000.0).unsqueeze(1)

        # 1. time (+ SDXL text_time) embedding
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            tdt = torch.float32 if isinstance(timestep, float) else torch.int64
            timesteps = torch.tensor([timesteps], dtype=tdt, device=sample.device)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None].to(sample.device)
        timesteps = timesteps.expand(sample.shape[0])
        t_emb = timestep_embedding(timesteps, cfg.block_out_channels[0], cfg.flip_sin_to_cos, cfg.freq_shift)
        emb = self.time_embedding(t_emb.to(dtype))
        if cfg.addition_embed_type == "text_time":
            if added_cond_kwargs is None or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
                raise ValueError("addition_embed_type 'text_time' requires `text_embeds` and `time_ids` in "
                                 "`added_cond_kwargs`")
            text_embeds, time_ids = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
            time_embeds = timestep_embedding(time_ids.flatten(), cfg.addition_time_embed_dim, cfg.flip_sin_to_cos,
                                             cfg.freq_shift).reshape(text_embeds.shape[0], -1)
            add_embeds = torch.cat([text_embeds.to(dtype), time_embeds.to(dtype)], dim=-1)
            emb = emb + self.add_embedding(add_embeds)

        encoder_hidden_states = encoder_hidden_states.to(dtype)
        if cfg.center_input_sample:
            sample = 2 * sample - 1.0

        # 2. down
        sample = sample.to(dtype)
        if getattr(self, "channels_last", False):
            sample = sample.contiguous(memory_format=torch.channels_last)
        sample = conv2d_bias(self.conv_in, sample)
        res_samples = (sample,)
        for block in self.down_blocks:
            sample, outs = block(sample, emb, encoder_hidden_states, encoder_attention_mask)
            res_samples += outs
        # 3. mid
        sample = self.mid_block(sample, emb, encoder_hidden_states, encoder_attention_mask)
        # 4. up
        for i, block in enumerate(self.up_blocks):
            n = len(block.resnets)
            res, res_samples = res_samples[-n:], res_samples[:-n]
            upsample_size = None
            if i != len(self.up_blocks) - 1 and forward_upsample_size:
                upsample_size = res_samples[-1].shape[2:]
            sample = block(sample, res, emb, encoder_hidden_states, encoder_attention_mask, upsample_size)
        # 5. out
        sample = self.conv_out(self.conv_norm_out(sample))
        if not return_dict:
            return (sample,)
        return UNet2DConditionOutput(sample=sample)
