"""GPU parity of the host-side stack (UNet / token decoder / scheduler / VAE / generate_image) against the oracle on
identical synthetic weights, tokens and explicit noise.  Product path = pigeon_amd modules
calling the HIP kernels through the C ABI; oracle = oracle.models_ref / pipeline_ref on the CPU.

Tolerances are FIXED constants, stated per test: relative L2 of the HIP bf16 result against the fp32 oracle (the
reference arithmetic on the unrounded synthetic weights, so the figure includes the weights' bf16 rounding), with the value measured on MI355X beside each bar (bar =
measured x 1.35-1.45).  Two bf16 evaluations of a deep network decorrelate at the bf16 noise floor (different GEMM
accumulation orders flip roundings, which then propagate), so the oracle's own bf16-autocast evaluation is printed next
to it for scale - the HIP path sits within 10 % of that evaluation's distance to fp32 in every test - but it no longer
sets the bar.  The same modules at real SDXL widths and in fp16 (8x lower floor): tests/test_gpu_widths.py."""
import numpy as np
import pytest
import torch

from oracle import models_ref, ops_ref, pipeline_ref, weights
from oracle.models_ref import Policy

pytestmark = pytest.mark.gpu
DEV = "cuda"

TINY_UNET = dict(block_out_channels=(64, 128, 128), attention_head_dim=(1, 2, 2), transformer_layers_per_block=(1, 1, 2),
                 cross_attention_dim=2048, layers_per_block=2, sample_size=16)
TINY_VAE = dict(block_out_channels=(32, 64), layers_per_block=1, latent_channels=4, norm_num_groups=32,
                scaling_factor=0.13025)
SCHED = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
             steps_offset=1, clip_sample=False, set_alpha_to_one=False, timestep_spacing="leading")


def _within(got, want_bf16, want_fp32, bar, what=""):
    e_hip = _rel(got, want_fp32)[1]
    e_ref = _rel(want_bf16, want_fp32)[1]
    print(f"{what} relL2 hip-vs-fp32 {e_hip:.4f} (bar {bar})  oracle_bf16-vs-fp32 {e_ref:.4f}  hip-vs-oracle_bf16 {_rel(got, want_bf16)[1]:.4f}")
    assert e_hip < bar, (what, e_hip, bar, e_ref)


def _rel(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-6)).item(), \
           ((got - want).norm() / want.norm().clamp_min(1e-6)).item()


@pytest.fixture(scope="module")
def stack():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pigeon_amd.pipeline import PigeonPixelDecoder
    from pigeon_amd.schedulers import DDIMScheduler
    from pigeon_amd.token_decoder import HighresVQDecoder
    from pigeon_amd.unet import UNet2DConditionModel
    from pigeon_amd.vae import AutoencoderKL
    unet = UNet2DConditionModel(**TINY_UNET)
    unet_sd = weights.synth_like(unet, 11)
    unet.load_state_dict(unet_sd)
    dec = HighresVQDecoder(depth=2)
    dec_sd = weights.synth_like(dec, 123)
    dec.load_state_dict(dec_sd)
    vae = AutoencoderKL(**TINY_VAE)
    vae_sd = weights.synth_like(vae, 5)
    vae.load_state_dict(vae_sd)
    codebook = weights.synth_codebook()
    pipe = PigeonPixelDecoder(codebook, dec, unet, vae, DDIMScheduler(**SCHED), "bf16", DEV)
    bf = lambda sd: {k: v.to(torch.bfloat16) for k, v in sd.items()}
    return dict(pipe=pipe, unet_cfg=dict(unet.config), unet_sd=bf(unet_sd), dec_sd=bf(dec_sd), vae_sd=bf(vae_sd),
                vae_cfg=dict(vae.config), codebook=codebook, unet_sd32=unet_sd, dec_sd32=dec_sd)


def test_native_library_is_the_one_in_tree(stack):
    from pigeon_amd import _lib
    assert _lib.load().pgn_version() == 112
    maps = open("/proc/self/maps").read()
    assert "pigeon_amd/lib/libpigeon_hip.so" in maps


def test_token_decoder_gpu_matches_oracle(stack):
    tokens = weights.synth_tokens(4, seed=3)
    pe, pooled = stack["pipe"].generate_image_embeds(tokens)
    emb, nums = ops_ref.token_gather_ref(tokens, stack["codebook"], torch.bfloat16)
    want_pe, want_pooled = models_ref.token_decoder_forward(Policy(torch.bfloat16), stack["dec_sd"], emb, nums)
    assert pe.shape == (4, 256, 2048) and pooled.shape == (4, 1280) and pe.dtype == torch.bfloat16
    emb32, _ = ops_ref.token_gather_ref(tokens, stack["codebook"], torch.float32)
    want32, want_pooled32 = models_ref.token_decoder_forward(Policy(torch.float32), stack["dec_sd32"], emb32, nums)
    _within(pe, want_pe, want32, 0.014, "token decoder prompt_embeds")      # measured 0.0102
    _within(pooled, want_pooled, want_pooled32, 0.010, "token decoder pooled")     # measured 0.0069


def _golden_rel(got, want):
    got, want = torch.as_tensor(got).double(), torch.as_tensor(want).double()
    return ((got - want).norm() / want.norm()).item()


@pytest.mark.parametrize("dt,bar", [(torch.float16, 0.002), (torch.bfloat16, 0.016)], ids=["fp16", "bf16"])
def test_product_token_decoders_match_the_reference_golden_vectors(golden_dir, dt, bar):
    # tests/golden/*.npz hold outputs of the reference's OWN modeling_decoder.py (HighresVQDecoder / VQDecoder, fp32,
    # scripts/gen_golden.py).  The product decoders run on the GPU through the HIP gather / attention / LayerNorm
    # kernels in 16-bit and are compared with those fp32 vectors under a FIXED bar (measured relL2, weights rounded
    # to the 16-bit type included: fp16 0.0009-0.0013, bf16 0.007-0.010): this is the one part of the path whose parity is pinned by the reference itself.
    import os
    from pigeon_amd import ops
    from pigeon_amd.token_decoder import HighresVQDecoder, VQDecoder
    codebook = weights.synth_codebook().to(DEV)
    g = np.load(os.path.join(golden_dir, "token_decoder_depth2.npz"))
    m = HighresVQDecoder(depth=2)
    m.load_state_dict(weights.synth_like(m, seed=123))
    m = m.to(DEV, dt).eval().requires_grad_(False)
    tokens = [t.to(DEV) for t in torch.from_numpy(g["tokens"])]
    out = ops.token_gather(tokens, codebook, out_dtype=dt)
    assert out["token_nums"].tolist() == g["token_nums"].tolist()
    rows = torch.from_numpy(g["prompt_rows"])
    with torch.no_grad():
        total = int(out["token_nums"].sum())
        for what, (prompt, pooled) in (("padded", m.forward_padded(out["padded"], out["mask"])),
                                       ("packed", m(out["packed"][:total], out["token_nums"])),
                                       ("fused token embed", m.forward_tokens(tokens, codebook))):
            e1 = _golden_rel(prompt[:, rows].float().cpu(), g["prompt_embeds_rows"])
            e2 = _golden_rel(pooled.float().cpu(), g["pooled"])
            print(f"HighresVQDecoder {what} {dt}: relL2 vs reference golden {e1:.5f} / pooled {e2:.5f}")
            assert e1 < bar and e2 < bar, (what, e1, e2)
    g = np.load(os.path.join(golden_dir, "vq_decoder_lowres_depth2.npz"))
    m = VQDecoder(depth=2)
    m.load_state_dict(weights.synth_like(m, seed=321))
    m = m.to(DEV, dt).eval().requires_grad_(False)
    tokens = [t.to(DEV) for t in torch.from_numpy(g["tokens"])]
    out = ops.token_gather(tokens, codebook, out_dtype=dt)
    rows = torch.from_numpy(g["rows"])
    with torch.no_grad():
        xrec = m.forward_padded(out["padded"], out["mask"])
    e = _golden_rel(xrec[:, rows].float().cpu(), g["xrec_rows"])
    print(f"VQDecoder {dt}: relL2 vs reference golden {e:.5f}")
    assert xrec.shape == (4, 256, 768) and e < bar
    with torch.no_grad():
        e = _golden_rel(m.forward_tokens(tokens, codebook)[:, rows].float().cpu(), g["xrec_rows"])
    print(f"VQDecoder fused token embed {dt}: relL2 vs reference golden {e:.5f}")
    assert e < bar


def _unet_inputs(B, h, w, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, h, w, generator=g)
    ctx = (torch.randn(B, 256, 2048, generator=g) * 0.6).to(torch.bfloat16)
    pooled = (torch.randn(B, 1280, generator=g) * 0.5).to(torch.bfloat16)
    ids = torch.tensor([[8.0 * w, 8.0 * h, 0, 0, 8.0 * w, 8.0 * h]], dtype=torch.bfloat16).repeat(B, 1)
    return x, ctx, pooled, ids


@pytest.mark.parametrize("hw", [(16, 16), (20, 12)])
def test_unet_forward_matches_oracle(stack, hw):
    h, w = hw
    x, ctx, pooled, ids = _unet_inputs(4, h, w)
    unet = stack["pipe"].unet
    got = unet(x.to(DEV), 961, encoder_hidden_states=ctx.to(DEV),
               added_cond_kwargs={"text_embeds": pooled.to(DEV), "time_ids": ids.to(DEV)}).sample
    assert got.dtype == torch.bfloat16 and got.shape == x.shape
    want = models_ref.unet_forward(Policy(torch.bfloat16), stack["unet_sd"], stack["unet_cfg"], x, torch.tensor(961),
                                   ctx, {"text_embeds": pooled, "time_ids": ids})
    want32 = models_ref.unet_forward(Policy(torch.float32), stack["unet_sd32"], stack["unet_cfg"], x, torch.tensor(961),
                                     ctx.float(), {"text_embeds": pooled.float(), "time_ids": ids.float()})
    _within(got, want, want32, 0.026, f"unet {hw}")    # measured 0.0169 / 0.0184 (16 resnets + 12 transformer blocks)


def test_unet_key_padding_mask_path(stack):
    # lowres-style call: encoder_attention_mask (lavit_for_pigeon.py:873-874) -> key bias inside the HIP attention
    x, ctx, pooled, ids = _unet_inputs(2, 16, 16, seed=4)
    mask = torch.ones(2, 256, dtype=torch.bool)
    mask[0, 77:] = False
    mask[1, 200:] = False
    unet = stack["pipe"].unet
    kw = {"text_embeds": pooled.to(DEV), "time_ids": ids.to(DEV)}
    got = unet(x.to(DEV), 500, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=kw,
               encoder_attention_mask=mask.to(DEV)).sample
    want = models_ref.unet_forward(Policy(torch.bfloat16), stack["unet_sd"], stack["unet_cfg"], x, torch.tensor(500), ctx,
                                   {"text_embeds": pooled, "time_ids": ids}, encoder_attention_mask=mask)
    want32 = models_ref.unet_forward(Policy(torch.float32), stack["unet_sd32"], stack["unet_cfg"], x, torch.tensor(500),
                                     ctx.float(), {"text_embeds": pooled.float(), "time_ids": ids.float()},
                                     encoder_attention_mask=mask)
    _within(got, want, want32, 0.026, "unet key-padding mask")    # measured 0.0193
    unmasked = unet(x.to(DEV), 500, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=kw).sample
    assert _rel(unmasked, want32)[1] > 2 * _rel(got, want32)[1]   # the mask matters


def test_scheduler_steps_on_gpu_match_oracle(stack):
    from pigeon_amd.schedulers import DDIMScheduler, EulerDiscreteScheduler
    g = torch.Generator().manual_seed(2)
    noise_pred = torch.randn(8, 4, 16, 16, generator=g).to(torch.bfloat16)
    lat = torch.randn(4, 4, 16, 16, generator=g)
    for extra in (dict(), dict(clip_sample=True, clip_sample_range=2.0), dict(prediction_type="v_prediction"),
                  dict(prediction_type="sample"), dict(set_alpha_to_one=True, timestep_spacing="trailing")):
        cfg = dict(SCHED)
        cfg.update(extra)
        s, r = DDIMScheduler(**cfg), pipeline_ref.DDIMRef(**cfg)
        s.set_timesteps(5, device=DEV)
        r.set_timesteps(5)
        x_g, x_r = lat.to(DEV), lat.clone()
        for t in s.timesteps:           # device timesteps, as the reference iterates them (:744)
            x_g = s.step_cfg(noise_pred.to(DEV), t, x_g, 7.0).prev_sample
            x_r = r.step(ops_ref.cfg_combine_ref(noise_pred, 7.0), int(t), x_r)[0]
            assert torch.equal(x_g.cpu(), x_r), extra
        # plain diffusers-style step(model_output, t, sample) with an already-guided prediction
        guided = ops_ref.cfg_combine_ref(noise_pred, 7.0)
        out = s.step(guided.to(DEV), s.timesteps[0], lat.to(DEV))
        assert torch.equal(out.prev_sample.cpu(), r.step(guided, int(s.timesteps[0]), lat)[0])
    ecfg = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading", steps_offset=1)
    s, r = EulerDiscreteScheduler(**ecfg), pipeline_ref.EulerRef(**ecfg)
    s.set_timesteps(4, device=DEV)
    r.set_timesteps(4)
    x_g, x_r = lat.to(DEV) * float(s.init_noise_sigma), lat * r.init_noise_sigma
    for t in s.timesteps:
        assert s.scale_divisor(t) == float((r.sigmas[r.idx] ** 2 + 1) ** 0.5)
        x_g = s.step_cfg(noise_pred.to(DEV), t, x_g, 5.0).prev_sample
        x_r = r.step(ops_ref.cfg_combine_ref(noise_pred, 5.0), t, x_r)[0]
        assert torch.equal(x_g.cpu(), x_r)


def test_ddim_thresholding_branch(stack):
    from pigeon_amd.schedulers import DDIMScheduler
    cfg = dict(SCHED, thresholding=True, dynamic_thresholding_ratio=0.9, sample_max_value=1.5, prediction_type="sample")
    s = DDIMScheduler(**cfg)
    s.set_timesteps(5)
    g = torch.Generator().manual_seed(3)
    x0 = (torch.randn(2, 4, 8, 8, generator=g) * 2).to(torch.bfloat16)
    lat = torch.randn(2, 4, 8, 8, generator=g)
    out = s.step(x0.to(DEV), s._host_timesteps[0], lat.to(DEV))
    thr = pipeline_ref.threshold_sample_ref(x0.float(), 0.9, 1.5)       # the oracle's own restatement (explicit sort)
    assert out.pred_original_sample.abs().max().item() <= 1.0 + 1e-6
    assert torch.allclose(out.pred_original_sample.cpu().float(), thr, atol=2e-6, rtol=0)
    # prev_sample from the thresholded x0 (prediction_type "sample": eps re-derived from x0), fp32 closed form
    t, n = int(s._host_timesteps[0]), 5
    r = pipeline_ref.DDIMRef(**cfg)
    r.set_timesteps(n)
    a_t, a_p = r.alphas_cumprod[t].double(), r.alphas_cumprod[t - 1000 // n].double()
    eps = (lat.double() - a_t.sqrt() * thr.double()) / (1 - a_t).sqrt()
    want_prev = a_p.sqrt() * thr.double() + (1 - a_p).sqrt() * eps
    assert torch.allclose(out.prev_sample.cpu().double(), want_prev, atol=2e-5, rtol=0)


def test_vae_decode_matches_oracle(stack):
    g = torch.Generator().manual_seed(6)
    z = torch.randn(2, 4, 16, 16, generator=g)
    pipe = stack["pipe"]
    got = pipe.vae.decode(z.to(DEV, torch.bfloat16) / 0.13025).sample
    want = models_ref.vae_decode(Policy(torch.bfloat16), stack["vae_sd"], stack["vae_cfg"], z.to(torch.bfloat16) / 0.13025)
    assert got.shape == (2, 3, 32, 32)
    want32 = models_ref.vae_decode(Policy(torch.float32), {k: v.float() for k, v in stack["vae_sd"].items()},
                                   stack["vae_cfg"], z.to(torch.bfloat16).float() / 0.13025)
    _within(got, want, want32, 0.010, "vae decode")    # measured 0.0068
    u8 = pipe.decode_latents(z.to(DEV), "uint8").cpu().numpy()
    pil = pipe.decode_latents(z.to(DEV), "pil")
    assert u8.shape == (2, 32, 32, 3) and u8.dtype == np.uint8 and len(pil) == 2
    assert np.array_equal(np.asarray(pil[1]), u8[1])
    ref_img = ((want.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy() * 255).round().astype("uint8")
    assert np.abs(u8.astype(int) - ref_img.astype(int)).max() <= 6   # 2 % of 255 on a bf16 decoder


def test_fp16_vae_is_upcast_for_decoding_and_cast_back(stack):
    # lavit_for_pigeon.py:766-778,796-813: fp16 VAE + force_upcast -> up blocks in fp32, post_quant_conv / conv_in /
    # mid block stay fp16, everything back to fp16 afterwards
    from pigeon_amd.pipeline import PigeonPixelDecoder
    from pigeon_amd.vae import AutoencoderKL
    vae = AutoencoderKL(**dict(TINY_VAE, force_upcast=True))
    sd = weights.synth_like(vae, 5)
    vae.load_state_dict(sd)
    from pigeon_amd.token_decoder import HighresVQDecoder
    from pigeon_amd.unet import UNet2DConditionModel
    p16 = PigeonPixelDecoder(stack["codebook"], HighresVQDecoder(depth=1), UNet2DConditionModel(**TINY_UNET), vae, None,
                             "fp16", DEV)              # own (throw-away) modules: the constructor casts in place
    if True:
        assert p16.vae.dtype == torch.float16
        z = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(6))
        seen = {}
        hook = p16.vae.decoder.up_blocks[0].register_forward_pre_hook(lambda m, a: seen.update(dt=a[0].dtype))
        u8 = p16.decode_latents(z.to(DEV), "uint8").cpu().numpy()
        hook.remove()
        assert seen["dt"] == torch.float32                                    # the up blocks ran in fp32
        assert all(p.dtype == torch.float16 for p in p16.vae.parameters())     # and were cast back (:777-778)
        h = lambda sd_: {k: v.to(torch.float16) for k, v in sd_.items()}
        want = models_ref.vae_decode(Policy(torch.float16), h(sd), dict(vae.config), z.to(torch.float16) / 0.13025,
                                     upscale_policy=Policy(torch.float32))
        ref_img = ((want.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy() * 255).round().astype("uint8")
        assert np.abs(u8.astype(int) - ref_img.astype(int)).max() <= 3        # fp32 up blocks: ~1 grey level


def test_safety_checker_hook_gets_unit_range_arrays_and_can_black_out(stack):
    # lavit_for_pigeon.py:784-785: images pass through the checker as float NHWC in [0, 1] before PIL conversion
    pipe = stack["pipe"]
    z = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(6)).to(DEV)
    plain = pipe.decode_latents(z, "uint8").cpu().numpy()
    seen = {}

    def checker(arr):
        seen["range"] = (float(arr.min()), float(arr.max()), arr.shape, arr.dtype)
        arr = arr.copy()
        arr[1] = 0.0
        return arr, [False, True]

    pipe.safety_checker = checker
    try:
        u8 = pipe.decode_latents(z, "uint8").cpu().numpy()
        pil = pipe.decode_latents(z, "pil")
    finally:
        pipe.safety_checker = None
    lo, hi, shape, dt = seen["range"]
    assert 0.0 <= lo and hi <= 1.0 and shape == (2, 32, 32, 3) and dt == np.float32
    assert np.array_equal(u8[0], plain[0]) and not u8[1].any() and not np.asarray(pil[1]).any()


@pytest.mark.parametrize("mode", ["highres", "lowres"])
def test_from_pretrained_reads_the_lavit_checkpoint_layout(tmp_path, mode):
    # lavit_for_pigeon.py:148,166-170,185-191 + modeling_decoder.py:340-351 + modeling_visual_tokenzier.py:437-442
    import json
    from pigeon_amd.pipeline import PigeonPixelDecoder
    from pigeon_amd.token_decoder import HighresVQDecoder, VQDecoder
    from pigeon_amd.unet import UNet2DConditionModel
    from pigeon_amd.vae import AutoencoderKL
    low = mode == "lowres"
    unet = UNet2DConditionModel(**(TINY_SD15 if low else TINY_UNET))
    vae = AutoencoderKL(**TINY_VAE)
    dec = (VQDecoder if low else HighresVQDecoder)(depth=1)
    root = tmp_path / "LaVIT"
    diff = root / ("pixel_decoding" if low else "highres_pixel_decoding")
    for sub in ("unet", "vae", "scheduler"):
        (diff / sub).mkdir(parents=True)
    (root / "visual_tokenizer").mkdir()
    sds = {"unet": weights.synth_like(unet, 1), "vae": weights.synth_like(vae, 2), "dec": weights.synth_like(dec, 3)}
    torch.save(sds["unet"], diff / "unet" / "diffusion_pytorch_model.bin")
    torch.save(sds["vae"], diff / "vae" / "diffusion_pytorch_model.bin")
    json.dump({k: v for k, v in unet.config.items()}, open(diff / "unet" / "config.json", "w"))
    json.dump({k: v for k, v in vae.config.items()}, open(diff / "vae" / "config.json", "w"))
    json.dump(dict(SCHED, _class_name="DDIMScheduler"), open(diff / "scheduler" / "scheduler_config.json", "w"))
    torch.save(sds["dec"], root / "visual_tokenizer" / ("tokenizer_decoder.bin" if low else "highres_tokenizer_decoder.bin"))
    codebook = weights.synth_codebook()
    torch.save({"quantize.embedding.weight": codebook, "encoder.something": torch.zeros(3)},
               root / "visual_tokenizer" / "tokenizer_encoder.bin")
    if low:
        torch.save(torch.randn(1, 77, 768), diff / "uncond_embeddings.bin")
    pipe = PigeonPixelDecoder.from_pretrained(str(root), pixel_decoding=mode, device=DEV, decoder_depth=1)
    assert pipe.pixel_decoding == mode and torch.equal(pipe.codebook.cpu(), codebook)
    assert torch.equal(pipe.unet.state_dict()["conv_in.weight"].cpu(), sds["unet"]["conv_in.weight"].to(torch.bfloat16))
    assert pipe.scheduler.config.steps_offset == 1 and pipe.vae.dtype == torch.bfloat16
    out = pipe.generate_image(image_tokens=weights.synth_tokens(1, seed=2), width=128, height=128, num_inference_steps=1,
                              guidance_scale_for_diffusion=3.0, generator=None if not low else torch.Generator().manual_seed(0))
    assert len(out) == (2 if low else 3) and out[0][0].size == (32, 32)


def test_generate_image_per_step_latents_match_oracle(stack):
    pipe = stack["pipe"]
    tokens = weights.synth_tokens(2, seed=9)
    k, steps, H, W = 2, 4, 128, 160        # latent 16 x 20, CFG batch 8
    g = torch.Generator().manual_seed(123)
    noise = torch.randn(2 * k, 4, H // 8, W // 8, generator=g)
    got_steps = []
    images, pe, pooled = pipe.generate_image(image_tokens=[t.to(DEV) for t in tokens], num_return_images=k, width=W,
                                             height=H, num_inference_steps=steps, guidance_scale_for_diffusion=7.0,
                                             latents=noise.to(DEV), output_type="uint8",
                                             callback=lambda i, t, x: got_steps.append(x.cpu().clone()))
    ref = pipeline_ref.generate_image_ref(Policy(torch.bfloat16), tokens, stack["codebook"], stack["dec_sd"],
                                          stack["unet_sd"], stack["unet_cfg"], pipeline_ref.DDIMRef(**SCHED), noise,
                                          num_return_images=k, width=W, height=H, num_inference_steps=steps,
                                          guidance_scale=7.0, vae_sd=stack["vae_sd"], vae_cfg=stack["vae_cfg"])
    assert len(got_steps) == steps and got_steps[0].dtype == torch.float32
    f32 = lambda sd: {k: v.float() for k, v in sd.items()}
    ref32 = pipeline_ref.generate_image_ref(Policy(torch.float32), tokens, stack["codebook"], stack["dec_sd32"],
                                            stack["unet_sd32"], stack["unet_cfg"], pipeline_ref.DDIMRef(**SCHED), noise,
                                            num_return_images=k, width=W, height=H, num_inference_steps=steps,
                                            guidance_scale=7.0)
    # per-step latents (eps noise x guidance 7 accumulates over the 4 guided steps)
    for i, (a, b, c) in enumerate(zip(got_steps, ref["latents_per_step"], ref32["latents_per_step"])):
        _within(a, b, c, 0.043, f"latents after step {i}")    # measured 0.0304-0.0308
    assert images.shape == (2 * k, H // 4, W // 4, 3)   # tiny VAE upsamples x2 once (16x20 -> 32x40)
    d = np.abs(images.cpu().numpy().astype(int) - ref["images"].astype(int))
    assert d.mean() < 3.0 and np.percentile(d, 99) <= 12, (d.mean(), d.max())
    # conditioning returned as the reference returns it (:789)
    assert pe.shape == (2, 256, 2048) and pooled.shape == (2, 1280)
    # num_return_images copies of one token sequence share conditioning but not noise
    assert not torch.equal(got_steps[-1][0], got_steps[-1][1])


def test_generate_image_argument_errors_and_generator_path(stack):
    pipe = stack["pipe"]
    with pytest.raises(ValueError):
        pipe.generate_image()
    with pytest.raises(ValueError):
        pipe.generate_image(image_tokens=weights.synth_tokens(1), width=128, height=128,
                            latents=torch.zeros(1, 4, 8, 8))
    gen = torch.Generator(device=DEV).manual_seed(123)   # inference.py:317
    toks = weights.synth_tokens(1, seed=1)
    a = pipe.generate_image(image_tokens=toks, width=128, height=128, num_inference_steps=2, generator=gen,
                            output_type="latent")[0]
    gen.manual_seed(123)
    b = pipe.generate_image(image_tokens=toks, width=128, height=128, num_inference_steps=2, generator=gen,
                            output_type="latent")[0]
    # same seed -> same noise; MIOpen / hipBLASLt may pick different algorithms on a first call, so compare closely
    # rather than bitwise
    assert a.dtype == torch.float32 and a.shape == (1, 4, 16, 16)
    assert _rel(a, b)[1] < 0.06      # bf16 noise floor of two evaluations (stream-K GEMMs are not run-to-run bitwise)
    gen.manual_seed(124)
    c = pipe.generate_image(image_tokens=toks, width=128, height=128, num_inference_steps=2, generator=gen,
                            output_type="latent")[0]
    assert _rel(a, c)[1] > 0.5       # a different seed is a different image


def test_zero_context_rows_shortcut_changes_nothing(stack):
    # CFG's unconditional half conditions on all-zero embeddings (lavit_for_pigeon.py:710-711): cross-attention of
    # those rows is exactly to_out's bias, so skipping them must leave their eps rows bit-identical and the
    # conditional rows equal up to GEMM tile selection (smaller M in the q / out projections)
    x, ctx, pooled, ids = _unet_inputs(4, 16, 16, seed=8)
    ctx[:2] = 0
    pooled[:2] = 0
    unet = stack["pipe"].unet
    kw = dict(encoder_hidden_states=ctx.to(DEV), added_cond_kwargs={"text_embeds": pooled.to(DEV), "time_ids": ids.to(DEV)})
    full = unet(x.to(DEV), 500, **kw).sample
    unet.set_zero_context_rows(2)
    try:
        skip = unet(x.to(DEV), 500, **kw).sample
        unet.enable_cross_kv_cache(True)
        skip_cached = [unet(x.to(DEV), 500, **kw).sample for _ in range(2)][1]   # second call reads the cache
    finally:
        unet.enable_cross_kv_cache(False)
        unet.set_zero_context_rows(0)
    assert torch.equal(skip[:2], full[:2])
    assert _rel(skip[2:], full[2:])[1] < 0.02 and _rel(skip_cached, skip)[1] < 0.02
    # and the pipeline switches it on for the loop and off afterwards
    pipe = stack["pipe"]
    assert pipe.skip_zero_context
    pipe.generate_image(image_tokens=weights.synth_tokens(1, seed=1), width=128, height=128, num_inference_steps=1,
                        output_type="latent")
    from pigeon_amd.attention_processor import HipAttnProcessor
    assert all(p.zero_ctx_rows == 0 for p in unet.attn_processors.values() if isinstance(p, HipAttnProcessor))


def test_cross_kv_reuse_inside_a_denoising_loop_changes_nothing(stack):
    # K / V of cross-attention are projections of encoder_hidden_states alone: generate_image computes them at the
    # first step and reuses them (PGN_REUSE_CROSS_KV); same GEMMs on the same inputs -> bit-identical latents, and
    # nothing stays cached after the call
    from pigeon_amd.attention_processor import HipAttnProcessor
    pipe = stack["pipe"]
    assert pipe.reuse_cross_kv
    kw = dict(width=128, height=128, num_inference_steps=3, guidance_scale_for_diffusion=7.0, output_type="latent")
    toks = weights.synth_tokens(2, seed=31)
    noise = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(6)).to(DEV)
    hits = []
    procs = [p for p in pipe.unet.attn_processors.values() if isinstance(p, HipAttnProcessor)]
    reused = pipe.generate_image(image_tokens=toks, latents=noise, callback=lambda i, t, x: hits.append(
        sum(p._kv_cache is not None for p in procs)), **kw)[0]
    assert hits[0] > 0 and hits == [hits[0]] * 3          # every cross-attention layer holds its K / V during the loop
    assert all(p._kv_cache is None and not p.cache_cross_kv for p in procs)
    pipe.reuse_cross_kv = False
    try:
        plain = pipe.generate_image(image_tokens=toks, latents=noise, **kw)[0]
    finally:
        pipe.reuse_cross_kv = True
    assert torch.equal(reused, plain)
    other = pipe.generate_image(image_tokens=weights.synth_tokens(2, seed=32), latents=noise, **kw)[0]
    assert _rel(other, reused)[1] > 0.05                   # a new call projects its own context


def test_fused_qkv_projection_equals_separate_projections(stack, monkeypatch):
    # self-attention runs to_q / to_k / to_v as one GEMM over the concatenated weights (HipAttnProcessor._self_qkv):
    # each element is the same dot product, so the result equals three separate GEMMs up to tile selection; the
    # fused weight follows the parameters (in-place writes and replaced storage both rebuild it)
    from pigeon_amd import attention_processor as ap
    attn = ap.Attention(128, heads=2, dim_head=64).to(DEV, torch.bfloat16)
    attn(torch.zeros(1, 8, 128, device=DEV, dtype=torch.bfloat16))
    assert not hasattr(attn, "_qkv_fused")        # trainable weights with autograd on: separate projections
    attn.requires_grad_(False)
    g = torch.Generator().manual_seed(21)
    hs = torch.randn(3, 80, 128, generator=g).to(DEV, torch.bfloat16)
    fused = attn(hs)
    assert attn._qkv_fused[1].shape == (384, 128)
    monkeypatch.setattr(ap, "_FUSE_QKV", False)
    separate = attn(hs)
    monkeypatch.setattr(ap, "_FUSE_QKV", True)
    assert _rel(fused, separate)[1] < 0.01
    with torch.no_grad():
        attn.to_k.weight.mul_(-0.5)
    first = attn._qkv_fused[1]
    changed = attn(hs)
    assert attn._qkv_fused[1] is not first and _rel(changed, fused)[1] > 0.1
    monkeypatch.setattr(ap, "_FUSE_QKV", False)
    assert _rel(changed, attn(hs))[1] < 0.01
    monkeypatch.setattr(ap, "_FUSE_QKV", True)
    assert "_qkv_fused" not in attn.state_dict() and len(attn.state_dict()) == 5
    # the whole UNet: fused and separate projections agree to the bf16 noise floor
    x, ctx, pooled, ids = _unet_inputs(2, 16, 16, seed=9)
    unet = stack["pipe"].unet
    kw = dict(encoder_hidden_states=ctx.to(DEV), added_cond_kwargs={"text_embeds": pooled.to(DEV), "time_ids": ids.to(DEV)})
    a = unet(x.to(DEV), 500, **kw).sample
    monkeypatch.setattr(ap, "_FUSE_QKV", False)
    b = unet(x.to(DEV), 500, **kw).sample
    assert _rel(a, b)[1] < 0.02


@pytest.mark.parametrize("switch", [("pigeon_amd.unet", "ACCUMULATE_ATTENTION"), ("pigeon_amd.unet", "ACCUMULATE_FF"),
                                    ("pigeon_amd.layers", "FUSE_CONV_BIAS"), ("pigeon_amd.layers", "SPLIT_CONCAT"),
                                    ("pigeon_amd.layers", "HIP_UPSAMPLE"), ("pigeon_amd.layers", "KEEP_DERIVED"),
                                    ("pigeon_amd.layers", "PRODUCER_STATS")])
def test_host_side_restructurings_agree_with_the_plain_order(stack, monkeypatch, switch):
    # every restructuring around the library GEMMs / convolutions (INTEGRATION.md switch table) is the same real
    # arithmetic in another order: switched off, the UNet and the VAE agree with the default build to the bf16 floor
    import importlib
    mod = importlib.import_module(switch[0])
    assert getattr(mod, switch[1]) is True
    x, ctx, pooled, ids = _unet_inputs(2, 16, 20, seed=13)
    pipe = stack["pipe"]
    kw = dict(encoder_hidden_states=ctx.to(DEV), added_cond_kwargs={"text_embeds": pooled.to(DEV), "time_ids": ids.to(DEV)})
    z = torch.randn(1, 4, 16, 20, generator=torch.Generator().manual_seed(3)).to(DEV, torch.bfloat16)
    a, va = pipe.unet(x.to(DEV), 300, **kw).sample, pipe.vae.decode(z).sample
    monkeypatch.setattr(mod, switch[1], False)
    b, vb = pipe.unet(x.to(DEV), 300, **kw).sample, pipe.vae.decode(z).sample
    assert _rel(a, b)[1] < 0.02 and _rel(va, vb)[1] < 0.02


@pytest.mark.parametrize("producer_stats", [False, True], ids=["default", "producer_stats"])
def test_hip_graph_replay_equals_eager_loop(stack, monkeypatch, producer_stats):
    # enable_hip_graph(): the whole denoising loop is captured once per (shape, schedule, guidance) and replayed;
    # same kernels, same order -> same latents as the eager loop up to the GEMMs' run-to-run noise floor.
    # producer_stats: the GroupNorm partial statistics that ride on tensors between launches (ops.gn_partials; by
    # default only above 32 MiB, i.e. never at this size) are allocated and consumed under capture as well
    if producer_stats:
        from pigeon_amd import ops as _ops
        monkeypatch.setattr(_ops, "GN_STATS_MIN_BYTES", 0)
    pipe = stack["pipe"]
    kw = dict(width=128, height=160, num_inference_steps=3, guidance_scale_for_diffusion=7.0, output_type="latent")
    toks_a, toks_b = weights.synth_tokens(2, seed=11), weights.synth_tokens(2, seed=12)
    g = torch.Generator().manual_seed(5)
    noise_a, noise_b = torch.randn(2, 4, 20, 16, generator=g), torch.randn(2, 4, 20, 16, generator=g)
    eager_a = pipe.generate_image(image_tokens=toks_a, latents=noise_a.to(DEV), **kw)[0]
    eager_b = pipe.generate_image(image_tokens=toks_b, latents=noise_b.to(DEV), **kw)[0]
    pipe.enable_hip_graph(True)
    try:
        graph_a = pipe.generate_image(image_tokens=toks_a, latents=noise_a.to(DEV), **kw)[0]      # capture + replay
        assert len(pipe._graphs) == 1
        graph_b = pipe.generate_image(image_tokens=toks_b, latents=noise_b.to(DEV), **kw)[0]      # replay, new inputs
        graph_a2 = pipe.generate_image(image_tokens=toks_a, latents=noise_a.to(DEV), **kw)[0]
        assert len(pipe._graphs) == 1
        steps = []
        pipe.generate_image(image_tokens=toks_a, latents=noise_a.to(DEV), callback=lambda i, t, x: steps.append(i), **kw)
        assert steps == [0, 1, 2]            # a per-step callback runs the eager loop
    finally:
        pipe.enable_hip_graph(False)
    assert not pipe._graphs
    assert _rel(graph_a, eager_a)[1] < 0.06 and _rel(graph_b, eager_b)[1] < 0.06 and _rel(graph_a2, graph_a)[1] < 0.06
    assert _rel(graph_a, eager_b)[1] > 0.5   # the replay really consumed the new inputs


def test_channels_last_and_nchw_modes_agree(stack):
    # PigeonPixelDecoder keeps activations channels_last by default; the NCHW mode runs the same modules through the
    # NCHW GroupNorm kernels.  Same weights, same input: equal up to the bf16 noise floor (different conv solvers).
    pipe = stack["pipe"]
    unet, vae = pipe.unet, pipe.vae
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 4, 16, 20, generator=g).to(DEV, torch.bfloat16)
    ctx = torch.randn(4, 256, 2048, generator=g).to(DEV, torch.bfloat16)
    added = {"text_embeds": torch.randn(4, 1280, generator=g).to(DEV, torch.bfloat16),
             "time_ids": torch.tensor([[160, 128, 0, 0, 160, 128]], device=DEV).repeat(4, 1)}
    z = torch.randn(2, 4, 16, 20, generator=g).to(DEV, torch.bfloat16)
    assert unet.channels_last and vae.channels_last
    try:
        a = unet(x, 500, encoder_hidden_states=ctx, added_cond_kwargs=added).sample
        va = vae.decode(z).sample
        unet.enable_channels_last(False)
        vae.enable_channels_last(False)
        b = unet(x, 500, encoder_hidden_states=ctx, added_cond_kwargs=added).sample
        vb = vae.decode(z).sample
    finally:
        unet.enable_channels_last(True)
        vae.enable_channels_last(True)
    assert b.is_contiguous()
    assert _rel(a, b)[1] < 0.02 and _rel(va, vb)[1] < 0.02


# ---- lowres decoder: pixel_decoding_origin (lavit_for_pigeon.py:822-895) -----------------------------------------
TINY_SD15 = dict(down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 block_out_channels=(64, 128, 128), attention_head_dim=2, transformer_layers_per_block=1,
                 cross_attention_dim=768, layers_per_block=1, use_linear_projection=False, addition_embed_type=None,
                 addition_time_embed_dim=None, projection_class_embeddings_input_dim=None, sample_size=16)
SD15_SCHED = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                  steps_offset=1, clip_sample=False, set_alpha_to_one=False, timestep_spacing="leading")


@pytest.fixture(scope="module")
def lowres_stack():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pigeon_amd.pipeline import PigeonPixelDecoder
    from pigeon_amd.schedulers import DDIMScheduler
    from pigeon_amd.token_decoder import VQDecoder
    from pigeon_amd.unet import UNet2DConditionModel
    from pigeon_amd.vae import AutoencoderKL
    unet = UNet2DConditionModel(**TINY_SD15)     # head dims 32 / 64 / 64 with 2 heads, conv proj_in / proj_out
    unet_sd = weights.synth_like(unet, 21)
    unet.load_state_dict(unet_sd)
    dec = VQDecoder(depth=2)
    dec_sd = weights.synth_like(dec, 321)
    dec.load_state_dict(dec_sd)
    vae = AutoencoderKL(**dict(TINY_VAE, scaling_factor=0.18215))
    vae_sd = weights.synth_like(vae, 5)
    vae.load_state_dict(vae_sd)
    codebook = weights.synth_codebook()
    uncond = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(77)) * 0.6   # the buffer of :168
    pipe = PigeonPixelDecoder(codebook, dec, unet, vae, DDIMScheduler(**SD15_SCHED), "bf16", DEV,
                              pixel_decoding="lowres", uncond_embeddings=uncond)
    bf = lambda sd: {k: v.to(torch.bfloat16) for k, v in sd.items()}
    return dict(pipe=pipe, unet_cfg=dict(unet.config), unet_sd=bf(unet_sd), dec_sd=bf(dec_sd), vae_sd=bf(vae_sd),
                vae_cfg=dict(vae.config), codebook=codebook, unet_sd32=unet_sd, dec_sd32=dec_sd, uncond=uncond)


def test_pixel_decoding_origin_per_step_latents_match_oracle(lowres_stack):
    st = lowres_stack
    pipe = st["pipe"]
    tokens = weights.synth_tokens(3, seed=5)
    steps, H, W = 4, 128, 160
    noise = torch.randn(3, 4, H // 8, W // 8, generator=torch.Generator().manual_seed(123))
    got_steps = []
    images, xrec = pipe.pixel_decoding_origin([t.to(DEV) for t in tokens], width=W, height=H,
                                              num_inference_steps=steps, guidance_scale_for_decoder=5.0, latents=noise,
                                              output_type="uint8",
                                              callback=lambda i, t, x: got_steps.append(x.cpu().clone()))
    kw = dict(width=W, height=H, num_inference_steps=steps, guidance_scale=5.0)
    ref = pipeline_ref.pixel_decoding_origin_ref(Policy(torch.bfloat16), tokens, st["codebook"], st["dec_sd"],
                                                 st["unet_sd"], st["unet_cfg"], pipeline_ref.DDIMRef(**SD15_SCHED),
                                                 st["uncond"], noise, vae_sd=st["vae_sd"], vae_cfg=st["vae_cfg"], **kw)
    ref32 = pipeline_ref.pixel_decoding_origin_ref(Policy(torch.float32), tokens, st["codebook"], st["dec_sd32"],
                                                   st["unet_sd32"], st["unet_cfg"], pipeline_ref.DDIMRef(**SD15_SCHED),
                                                   st["uncond"], noise, **kw)
    assert xrec.shape == (3, 256, 768) and xrec.dtype == torch.bfloat16
    _within(xrec, ref["xrec"], ref32["xrec"], 0.014, "lowres xrec")    # measured 0.0101
    assert len(got_steps) == steps
    for i, (a, b, c) in enumerate(zip(got_steps, ref["latents_per_step"], ref32["latents_per_step"])):
        _within(a, b, c, 0.028, f"lowres latents after step {i}")    # measured 0.0196-0.0206
    d = np.abs(images.cpu().numpy().astype(int) - ref["images"].astype(int))
    assert images.shape == (3, H // 4, W // 4, 3) and d.mean() < 3.0 and np.percentile(d, 99) <= 12, (d.mean(), d.max())


def test_lowres_facade_dispatch_and_host_noise(lowres_stack):
    # generate_image hands over to pixel_decoding_origin (:702-703); noise comes from a HOST generator (:858-860),
    # so a seeded run is reproducible from the CPU stream alone and equals an explicit-latents run
    pipe = lowres_stack["pipe"]
    toks = weights.synth_tokens(2, seed=8)
    out = pipe.generate_image(image_tokens=toks, width=128, height=128, num_inference_steps=2,
                              guidance_scale_for_diffusion=5.0, generator=torch.Generator().manual_seed(123),
                              output_type="latent")
    assert len(out) == 2 and out[0].shape == (2, 4, 16, 16) and out[1].shape == (2, 256, 768)
    noise = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(123))
    b = pipe.pixel_decoding_origin(toks, 128, 128, 2, 5.0, latents=noise, output_type="latent")[0]
    assert _rel(out[0], b)[1] < 0.06
    pil, _ = pipe.pixel_decoding_origin(toks, 128, 128, 1, 5.0, generator=torch.Generator().manual_seed(1))
    assert len(pil) == 2 and pil[0].size == (32, 32)
    from pigeon_amd.pipeline import PigeonPixelDecoder
    with pytest.raises(ValueError):
        PigeonPixelDecoder(lowres_stack["codebook"], pipe.tokenizer_decoder, pipe.unet, None, None, "bf16", DEV,
                           pixel_decoding="lowres")
