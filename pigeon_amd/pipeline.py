"""The hot path: visual tokens -> images.  Host-side mirror of `LaVITforPigeon.generate_image`
(Pigeon/models/lavit_for_pigeon.py:674-789, highres branch) with the same arguments, ordering, dtypes and
3-tuple return, driving the HIP kernels:

    generate_image_embeds (:897-920)   -> ops.token_gather (one fused launch pair) + HighresVQDecoder.forward_padded
    cat([latents]*2) + scale_model_input (:747-748) -> ops.cfg_expand_scale
    unet(...) (:752-757)               -> pigeon_amd.unet.UNet2DConditionModel (HIP attention + GroupNorm/SiLU)
    chunk(2) + guidance + scheduler.step (:760-764) -> scheduler.step_cfg (one fused launch)
    vae.decode + post-process (:766-787)            -> pigeon_amd.vae.AutoencoderKL + on-device uint8 conversion

Reference quirks kept on purpose: `target_size=(width, height)` (:724), all-zero unconditional embeddings including
the pooled one (:710-711), CFG batch order [neg...; pos...] followed by repeat_interleave (:731-736), fp32 latents.
"""
from typing import Callable, List, Optional, Sequence, Tuple, Union

import os

import numpy as np
import torch

from . import ops
from .schedulers import DDIMScheduler

# inference.py:58-65 — (height, width) per aspect-ratio flag
RATIO_DICT = {
    "1:1": (1024, 1024), "4:3": (896, 1152), "3:2": (832, 1216), "16:9": (768, 1344), "2:3": (1216, 832),
    "3:4": (1152, 896),
}


def numpy_to_pil(images: np.ndarray):
    """lavit_for_pigeon.py:922-936 (float [0,1] NHWC or already-uint8 NHWC -> list of PIL images)."""
    from PIL import Image
    if images.ndim == 3:
        images = images[None, ...]
    if images.dtype != np.uint8:
        images = (images * 255).round().astype("uint8")
    if images.shape[-1] == 1:
        return [Image.fromarray(image.squeeze(), mode="L") for image in images]
    return [Image.fromarray(image) for image in images]


# gather + in_proj + padding mask + pos_embed + key bias in one fused launch pair (ops.token_embed); "0" = the gather
# kernel followed by the decoder's own GEMM and elementwise launches
FUSE_TOKEN_EMBED = os.environ.get("PGN_FUSE_TOKEN_EMBED", "1") != "0"

class PigeonPixelDecoder:
    """Owns the frozen decoder stack (codebook, token decoder, UNet, scheduler, VAE) on one GPU."""

    MAX_GRAPHS = 2      # captured denoising loops kept (least recently used dropped first)

    def __init__(self, codebook: torch.Tensor, tokenizer_decoder, unet, vae, scheduler=None,
                 model_dtype: str = "bf16", device: Union[str, torch.device] = "cuda", channels_last: bool = True,
                 pixel_decoding: str = "highres", uncond_embeddings: Optional[torch.Tensor] = None,
                 safety_checker: Optional[Callable] = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PigeonPixelDecoder runs on an MI355X (device 'cuda:N'); there is no CPU path")
        self.dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(model_dtype, torch.float16)  # :229-236
        self.codebook = codebook.detach().to(self.device, torch.float32).contiguous()
        self.tokenizer_decoder = tokenizer_decoder.to(self.device, self.dtype).eval().requires_grad_(False)
        self.unet = unet.to(self.device, self.dtype).eval().requires_grad_(False)
        self.vae = None if vae is None else vae.to(self.device, self.dtype).eval().requires_grad_(False)
        self.scheduler = scheduler if scheduler is not None else DDIMScheduler()
        # check_safety (lavit_for_pigeon.py:58,206-212,784-785): callable(images float NHWC [0, 1]) -> (images, flags);
        # the CLIP-based StableDiffusionSafetyChecker itself is outside the accelerated path (off in run_inf.sh)
        self.safety_checker = safety_checker
        if pixel_decoding not in ("highres", "lowres"):
            raise ValueError("pixel_decoding is 'highres' or 'lowres'")
        self.pixel_decoding = pixel_decoding                       # lavit_for_pigeon.py:163
        # cross-attention skips the all-zero unconditional rows (see HipAttnProcessor.zero_ctx_rows)
        self.skip_zero_context = os.environ.get("PGN_SKIP_ZERO_CONTEXT", "1") != "0"
        self.use_hip_graph = os.environ.get("PGN_HIP_GRAPH", "0") == "1"      # see enable_hip_graph
        # cross-attention K / V are projections of encoder_hidden_states alone, which the loop never changes: compute
        # them at the first denoising step of a call and reuse them for the others (dropped when the call returns)
        self.reuse_cross_kv = os.environ.get("PGN_REUSE_CROSS_KV", "1") != "0"
        self._graphs = {}
        if pixel_decoding == "lowres":
            if uncond_embeddings is None:
                raise ValueError("pixel_decoding='lowres' needs the `uncond_embeddings` buffer "
                                 "(pixel_decoding/uncond_embeddings.bin, lavit_for_pigeon.py:168)")
            self.uncond_embeddings = uncond_embeddings.detach().to(self.device)
        # version-keyed tuning artefacts (library kernels): both calls WARN when the running library
        # builds do not match what ships in-tree; the flags go into bench.py's line
        from . import enable_tuned_gemms, warn_if_miopen_db_missed
        self.tuned_gemms = enable_tuned_gemms()
        self.miopen_db_hit = warn_if_miopen_db_missed()
        if channels_last:
            for m in (self.unet, self.vae):
                if m is not None and hasattr(m, "enable_channels_last"):
                    m.enable_channels_last(True)

    @classmethod
    def from_pretrained(cls, model_path: str, pixel_decoding: str = "highres", model_dtype: str = "bf16",
                        device: Union[str, torch.device] = "cuda", decoder_depth: int = 12, **kwargs):
        """Build the stack from a LaVIT checkpoint directory exactly as the reference does: codebook from
        `visual_tokenizer/tokenizer_encoder.bin` (`quantize.embedding.weight`, modeling_visual_tokenzier.py:437-442),
        token decoder by `build_tokenizer_decoder` (modeling_decoder.py:340-351), and the diffusion modules of
        `prepare_dm_to_generate_image` (lavit_for_pigeon.py:166-170,185-191): `{highres_,}pixel_decoding/{vae,scheduler,
        unet}` (+ `uncond_embeddings.bin` for the lowres decoder).  Pickle files go through weights_only loading."""
        import os
        from .token_decoder import build_tokenizer_decoder
        from .unet import UNet2DConditionModel
        from .vae import AutoencoderKL
        dtype = torch.bfloat16 if model_dtype == "bf16" else torch.float16
        enc_path = os.path.join(model_path, "visual_tokenizer", "tokenizer_encoder.bin")
        try:       # the file also holds the 1 B-parameter EVA-CLIP encoder: map it, copy the 2 MiB codebook out
            enc = torch.load(enc_path, map_location="cpu", weights_only=True, mmap=True)
        except (RuntimeError, ValueError):     # legacy (non-zip) serialisation cannot be mapped
            enc = torch.load(enc_path, map_location="cpu", weights_only=True)
        codebook = enc["quantize.embedding.weight"].float().clone()
        del enc
        decoder = build_tokenizer_decoder(model_path, pixel_decoding=pixel_decoding, depth=decoder_depth)
        diff_dir = os.path.join(model_path, "pixel_decoding" if pixel_decoding == "lowres" else "highres_pixel_decoding")
        vae = AutoencoderKL.from_pretrained(diff_dir, subfolder="vae", torch_dtype=dtype)
        scheduler = DDIMScheduler.from_pretrained(diff_dir, subfolder="scheduler")
        unet = UNet2DConditionModel.from_pretrained(diff_dir, subfolder="unet", use_safetensors=False, torch_dtype=dtype)
        uncond = None
        if pixel_decoding == "lowres":
            uncond = torch.load(os.path.join(diff_dir, "uncond_embeddings.bin"), map_location="cpu", weights_only=True)
        return cls(codebook, decoder, unet, vae, scheduler, model_dtype, device, pixel_decoding=pixel_decoding,
                   uncond_embeddings=uncond, **kwargs)

    # ---- :897-920 ------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_image_embeds(self, image_tokens: Union[Sequence[torch.Tensor], torch.Tensor]):
        """-> (prompt_embeds, pooled) from the highres decoder, `xrec` [B, 256, 768] from the lowres one."""
        toks = image_tokens if isinstance(image_tokens, torch.Tensor) else [t.to(self.device) for t in image_tokens]
        if FUSE_TOKEN_EMBED and self.tokenizer_decoder.in_proj.weight.dtype in (torch.bfloat16, torch.float16):
            return self.tokenizer_decoder.forward_tokens(toks, self.codebook)
        g = ops.token_gather(toks, self.codebook, out_dtype=self.dtype, packed=False, padded=True)
        return self.tokenizer_decoder.forward_padded(g["padded"], g["mask"])

    @staticmethod
    def _get_add_time_ids(original_size, crops_coords_top_left, target_size, dtype):
        return torch.tensor([list(original_size + crops_coords_top_left + target_size)], dtype=dtype)  # :791-794

    # ---- :674-789 ------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_image(self, batch=None, image_tokens=None, mask_ratio=0.2, hist_mask_ratio=0.2, add_special=True,
                       with_mask=True, llama_generation_config=None, num_return_images: int = 1, width: int = 512,
                       height: int = 512, original_size: Optional[Tuple[int, int]] = None,
                       crops_coords_top_left: Optional[Tuple[int, int]] = None, num_inference_steps: int = 25,
                       guidance_scale_for_diffusion: float = 7.0, generator: Optional[torch.Generator] = None,
                       latents: Optional[torch.Tensor] = None, output_type: str = "pil",
                       callback: Optional[Callable[[int, int, torch.Tensor], None]] = None):
        """Same contract as the reference plus three keyword extensions: `latents` (explicit initial noise, for
        reproducible parity runs and rank sharding), `output_type` in {"pil", "np", "uint8", "latent"} and a
        per-step `callback(i, t, latents)`."""
        if batch is None and image_tokens is None:
            raise ValueError("Generated image requires `batch` input or `image_tokens` input.")
        if image_tokens is None:
            raise NotImplementedError("token generation (the LLaMA stage) is outside the accelerated path: pass "
                                      "`image_tokens` as inference.py:389-399 does")
        if self.pixel_decoding == "lowres":                        # :702-703 (same positional hand-over)
            return self.pixel_decoding_origin(image_tokens, width, height, num_inference_steps,
                                              guidance_scale_for_diffusion, generator, latents=latents,
                                              output_type=output_type, callback=callback)
        prompt_embeds, pooled_prompt_embeds = self.generate_image_embeds(image_tokens)
        negative_prompt_embeds = torch.zeros_like(prompt_embeds)
        negative_pooled_prompt_embeds = torch.zeros_like(pooled_prompt_embeds)

        batch_size = len(prompt_embeds) * num_return_images
        shape = (batch_size, self.unet.config.in_channels, height // 8, width // 8)
        if latents is None:
            latents = torch.randn(shape, generator=generator, device=self.device)
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"latents must have shape {shape}")
            latents = latents.to(self.device, torch.float32)
        latents = latents * self.scheduler.init_noise_sigma
        self.scheduler.set_timesteps(num_inference_steps, device=self.device)

        target_size = (width, height)
        original_size = original_size if original_size is not None else target_size
        crops_coords_top_left = crops_coords_top_left if crops_coords_top_left is not None else (0, 0)
        add_time_ids = self._get_add_time_ids(original_size, crops_coords_top_left, target_size, prompt_embeds.dtype)
        add_time_ids = add_time_ids.repeat(batch_size, 1)
        input_embeds = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)
        add_text_embeds = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], dim=0)
        add_time_ids = torch.cat([add_time_ids, add_time_ids], dim=0).to(self.device)
        input_embeds = input_embeds.repeat_interleave(num_return_images, dim=0)
        add_text_embeds = add_text_embeds.repeat_interleave(num_return_images, dim=0)
        assert input_embeds.shape[0] == add_text_embeds.shape[0] == add_time_ids.shape[0] == latents.shape[0] * 2

        added_cond_kwargs = {"text_embeds": add_text_embeds, "time_ids": add_time_ids}
        from . import warn_if_untuned
        warn_if_untuned(2 * batch_size, height // 8, width // 8)
        host_timesteps = list(self.scheduler._host_timesteps)  # no device->host sync inside the loop
        # rows [0, batch_size) of input_embeds are the all-zero unconditional embeddings built above (:710-711, :731)
        zero_rows = batch_size if (self.skip_zero_context and hasattr(self.unet, "set_zero_context_rows")) else 0
        if zero_rows:
            self.unet.set_zero_context_rows(zero_rows)
        kv_before = self._begin_cross_kv_reuse()
        try:
            if self.use_hip_graph and callback is None:
                latents = self._denoise_graphed(latents, input_embeds, added_cond_kwargs, host_timesteps,
                                                guidance_scale_for_diffusion, zero_rows)
            else:
                for i, t in enumerate(host_timesteps):
                    latents = self._denoise_step(latents, t, t, input_embeds, added_cond_kwargs,
                                                 guidance_scale_for_diffusion)
                    if callback is not None:
                        callback(i, t, latents)
        finally:
            self._end_cross_kv_reuse(kv_before)
            if zero_rows:
                self.unet.set_zero_context_rows(0)

        if output_type == "latent":
            return latents, prompt_embeds, pooled_prompt_embeds
        images = self.decode_latents(latents, output_type)
        return images, prompt_embeds, pooled_prompt_embeds

    def _begin_cross_kv_reuse(self):
        """Switch the UNet's processors to K / V reuse for one denoising loop; returns what _end_cross_kv_reuse needs."""
        if not (self.reuse_cross_kv and hasattr(self.unet, "enable_cross_kv_cache")):
            return None
        before = bool(getattr(self.unet, "cross_kv_cache_enabled", False))
        self.unet.enable_cross_kv_cache(True)          # also drops anything cached by an earlier call
        return before

    def _end_cross_kv_reuse(self, before):
        if before is not None:
            self.unet.enable_cross_kv_cache(before)    # resets every processor's cache: nothing outlives the call

    def _denoise_step(self, latents, t_host, t_unet, input_embeds, added_cond_kwargs, guidance):
        """One iteration of :744-764: three fused launches around the UNet forward."""
        latent_model_input = ops.cfg_expand_scale(latents, self.unet.dtype, self.scheduler.scale_divisor(t_host))
        noise_pred = self.unet(latent_model_input, t_unet, encoder_hidden_states=input_embeds,
                               added_cond_kwargs=added_cond_kwargs).sample
        return self.scheduler.step_cfg(noise_pred, t_host, latents, guidance).prev_sample

    # ---- HIP graph of the whole denoising loop -------------------------------------------------------------------
    def enable_hip_graph(self, enabled: bool = True):
        """Capture the denoising loop (all steps: ~3 k launches each) into ONE hipGraph per (batch shape, schedule,
        guidance) and replay it.  A UNet forward takes ~34 ms of host time to enqueue; at UNet batch 32 the GPU needs
        250 ms for it and the host is hidden, at batch 2 (one image) the GPU needs ~20 ms and the loop is launch-bound:
        that is where the graph pays.  Off by default; per-step callbacks fall back to the eager loop."""
        self.use_hip_graph = bool(enabled)
        if not enabled:
            self._graphs.clear()
        return self

    def _denoise_graphed(self, latents, input_embeds, added_cond_kwargs, host_timesteps, guidance, zero_rows):
        # everything the capture bakes in: shapes, the schedule, guidance, the scheduler OBJECT and its configuration
        # (DDIM / Euler coefficients are launch constants), the UNet's dtype / device and parameter identity
        sched_cfg = getattr(self.scheduler, "config", None)
        key = (tuple(latents.shape), tuple(input_embeds.shape), tuple(host_timesteps), float(guidance), zero_rows,
               type(self.scheduler).__name__, id(self.scheduler),
               repr(sorted(sched_cfg.items())) if hasattr(sched_cfg, "items") else None,
               str(self.unet.dtype), str(self.unet.device), self.unet.conv_in.weight.data_ptr())
        entry = self._graphs.pop(key, None)           # re-inserted below: dict order = least recently used first
        if entry is None:
            st = {"lat": latents.clone(), "emb": input_embeds.clone(),
                  "text": added_cond_kwargs["text_embeds"].clone(), "ids": added_cond_kwargs["time_ids"].clone()}
            added = {"text_embeds": st["text"], "time_ids": st["ids"]}
            # timesteps as device tensors made BEFORE capture (a host->device copy cannot be captured)
            t_dev = [torch.tensor([t], device=self.device,
                                  dtype=torch.float32 if isinstance(t, float) else torch.int64) for t in host_timesteps]
            # one eager iteration on a side stream first: library handles, workspaces and MIOpen / hipBLASLt solution
            # look-ups must exist before the capture starts
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                self._denoise_step(st["lat"], host_timesteps[0], t_dev[0], st["emb"], added, guidance)
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._reset_scheduler_cursor()
            if getattr(self.unet, "cross_kv_cache_enabled", False):
                self.unet.enable_cross_kv_cache(True)   # drop the warm-up's K / V: the capture must hold its own
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                x = st["lat"]
                for t, td in zip(host_timesteps, t_dev):
                    x = self._denoise_step(x, t, td, st["emb"], added, guidance)
                st["out"] = x
            entry = (graph, st, t_dev)
        self._graphs[key] = entry
        while len(self._graphs) > self.MAX_GRAPHS:    # each entry pins a full activation pool
            self._graphs.pop(next(iter(self._graphs)))
        graph, st, _ = entry
        st["lat"].copy_(latents)
        st["emb"].copy_(input_embeds)
        st["text"].copy_(added_cond_kwargs["text_embeds"])
        st["ids"].copy_(added_cond_kwargs["time_ids"])
        graph.replay()
        return st["out"].clone()

    def _reset_scheduler_cursor(self):
        # schedulers with a step cursor (Euler) advanced it during the warm-up iteration
        if getattr(self.scheduler, "_step_index", None) is not None:
            self.scheduler._step_index = None

    # ---- :822-895 ------------------------------------------------------------------------------
    @torch.no_grad()
    def pixel_decoding_origin(self, image_tokens, width: int = 512, height: int = 512, num_inference_steps: int = 50,
                              guidance_scale_for_decoder: float = 5.0, generator: Optional[torch.Generator] = None,
                              latents: Optional[torch.Tensor] = None, output_type: str = "pil",
                              callback: Optional[Callable[[int, int, torch.Tensor], None]] = None):
        """The original LaVIT pixel decoder (SD-1.5 UNet): learned unconditional embeddings in the first rows of an
        otherwise zero context, boolean key masks for both CFG halves, noise drawn on the HOST (:858-860 has no
        `device=`, so `generator` is a CPU generator here) and a 2-tuple return `(images, xrec)`.  Keyword extensions
        as in `generate_image`."""
        if self.pixel_decoding != "lowres":
            raise RuntimeError("pixel_decoding_origin needs pixel_decoding='lowres' (VQDecoder + uncond_embeddings)")
        xrec = self.generate_image_embeds(image_tokens)                                        # :837
        batch_size, num_tokens, C = xrec.shape
        uncond = self.uncond_embeddings[0].to(xrec.dtype)                                      # :844
        n_u = len(uncond)
        encoder_hidden_uncond = torch.zeros(batch_size, num_tokens, C, dtype=xrec.dtype, device=self.device)
        encoder_hidden_uncond[:, :n_u] = uncond                                                # :845
        encoder_mask = torch.ones(batch_size, num_tokens, dtype=torch.bool, device=self.device)
        uncond_encoder_mask = torch.zeros(batch_size, num_tokens, dtype=torch.bool, device=self.device)
        uncond_encoder_mask[:, :n_u] = True                                                    # :850
        text_embeddings = torch.cat([encoder_hidden_uncond, xrec])                             # :855
        text_embeddings_mask = torch.cat([uncond_encoder_mask, encoder_mask])                  # :856

        shape = (batch_size, self.unet.config.in_channels, height // 8, width // 8)
        if latents is None:
            latents = torch.randn(shape, generator=generator)                                  # host noise, :858-860
        elif tuple(latents.shape) != shape:
            raise ValueError(f"latents must have shape {shape}")
        latents = (latents.float() * self.scheduler.init_noise_sigma).to(self.device)          # :861-862
        self.scheduler.set_timesteps(num_inference_steps, device=self.device)                  # :864
        kv_before = self._begin_cross_kv_reuse()
        try:
            for i, t in enumerate(list(self.scheduler._host_timesteps)):                       # :866
                latent_model_input = ops.cfg_expand_scale(latents, self.unet.dtype, self.scheduler.scale_divisor(t))
                noise_pred = self.unet(latent_model_input, t, encoder_hidden_states=text_embeddings,
                                       encoder_attention_mask=text_embeddings_mask).sample      # :873-874
                latents = self.scheduler.step_cfg(noise_pred, t, latents, guidance_scale_for_decoder).prev_sample
                if callback is not None:
                    callback(i, t, latents)
        finally:
            self._end_cross_kv_reuse(kv_before)
        if output_type == "latent":
            return latents, xrec
        # :883-884 run inside the autocast region: the division is fp32, the first convolution casts to bf16
        return self.decode_latents(latents, output_type, divide_in_fp32=True), xrec

    # ---- :766-787 ------------------------------------------------------------------------------
    @torch.no_grad()
    def decode_latents(self, latents: torch.Tensor, output_type: str = "pil", chunk: int = 1,
                       divide_in_fp32: bool = False):
        """VAE decode + post-process.  Images are decoded `chunk` at a time: samples are independent in the decoder,
        and a 16 x 256 x 1024 x 1024 activation exceeds the 2^31-element limit of torch's nearest-upsample launch."""
        if self.vae is None:
            raise RuntimeError("no VAE attached")
        # :766-770 — an fp16 VAE with force_upcast overflows: decode with the up blocks in fp32, cast back afterwards
        needs_upcasting = self.vae.dtype == torch.float16 and bool(self.vae.config.get("force_upcast", False))
        if needs_upcasting:
            self.upcast_vae()
        try:
            return self._decode_latents(latents, output_type, chunk, divide_in_fp32)
        finally:
            if needs_upcasting:
                self.vae.to(dtype=torch.float16)                                               # :777-778

    def upcast_vae(self):
        """lavit_for_pigeon.py:796-813: the VAE goes to fp32; with a fused attention processor (ours is one, like
        AttnProcessor2_0 / XFormersAttnProcessor there) post_quant_conv, conv_in and the mid block return to fp16."""
        dtype = self.vae.dtype
        self.vae.to(dtype=torch.float32)
        self.vae.post_quant_conv.to(dtype)
        self.vae.decoder.conv_in.to(dtype)
        self.vae.decoder.mid_block.to(dtype)

    def _decode_latents(self, latents, output_type, chunk, divide_in_fp32):
        vae_dtype = next(iter(self.vae.post_quant_conv.parameters())).dtype
        if divide_in_fp32:
            z = (latents / self.vae.config.scaling_factor).to(vae_dtype)
        else:
            z = latents.to(vae_dtype) / self.vae.config.scaling_factor                         # :773-774
        outs = []
        for i in range(0, z.shape[0], max(1, chunk)):
            image = self.vae.decode(z[i:i + chunk]).sample
            if output_type != "np":
                # uint8 NHWC on the device in one launch: 4x fewer bytes over PCIe than the reference's fp32 copy;
                # same arithmetic and rounding as :780-782 + numpy_to_pil ((x * 255).round(), half to even)
                image = ops.image_to_uint8(image)
            else:
                image = (image.float() / 2 + 0.5).clamp(0, 1)
            outs.append(image)
        image = torch.cat(outs, dim=0)
        if self.safety_checker is not None:
            # :784-785 / :815-820 — the reference hands float NHWC arrays in [0, 1] to its StableDiffusionSafetyChecker,
            # which blacks out flagged images; here the checker is a caller-supplied callable with that contract
            arr = image.cpu().permute(0, 2, 3, 1).numpy() if output_type == "np" else \
                image.cpu().numpy().astype(np.float32) / 255.0
            arr, _ = self.safety_checker(arr)
            if output_type == "np":
                return arr
            u8 = torch.from_numpy((np.asarray(arr) * 255).round().astype("uint8"))
            return u8.to(image.device) if output_type == "uint8" else numpy_to_pil(u8.numpy())
        if output_type == "np":
            return image.cpu().permute(0, 2, 3, 1).numpy()
        u8 = image
        if output_type == "uint8":
            return u8
        return numpy_to_pil(u8.cpu().numpy())
