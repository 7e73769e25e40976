"""BASELINE.json configs at their FULL sizes on the MI355X, checked through size-independent properties (the CPU oracle
needs minutes per UNet forward at these sizes; its parity is established on the small shapes of test_gpu_models.py and
test_gpu_kernels.py):

  configs[1]  batch 16 @ 1024x1024          sample independence (what rank sharding relies on), CFG identity
  configs[2]  1216x832 (ragged N 3952/988)   same, on the ragged attention / GroupNorm shapes, 50-step schedule
  configs[3]  8 images per (user, target)    repeat_interleave ordering of conditioning vs noise
  configs[4]  2x latent side, long cross-attention key sequences   attention properties at N = 16384, Nk = 1024
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def sdxl():
    """The full random-init SDXL-shaped stack of bench.py (UNet 2.57 B, token decoder 414 M, VAE), built once."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pigeon_amd import synthetic
    return synthetic.build_random_stack(DEV, "bf16", seed=123)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-9)).item()


def _tokens(n, seed):
    from pigeon_amd import synthetic
    return [t.to(DEV) for t in synthetic.synthetic_tokens(n, seed=seed)]


def test_config2_samples_are_independent_of_their_batch(sdxl):
    # Sharding contract: a sample's latents do not depend on which other samples share its batch, so an
    # N-GPU run reproduces the 1-GPU run.  Batch of 4 vs the same samples run 2 + 2 with the same noise rows; the only
    # difference allowed is GEMM/conv algorithm choice per batch shape (bf16 noise floor).
    toks = _tokens(4, 7)
    noise = torch.randn(4, 4, 128, 128, generator=torch.Generator().manual_seed(123)).to(DEV)
    kw = dict(width=1024, height=1024, num_inference_steps=2, guidance_scale_for_diffusion=7.0, output_type="latent")
    full = sdxl.generate_image(image_tokens=toks, latents=noise, **kw)[0]
    lo = sdxl.generate_image(image_tokens=toks[:2], latents=noise[:2], **kw)[0]
    hi = sdxl.generate_image(image_tokens=toks[2:], latents=noise[2:], **kw)[0]
    assert full.shape == (4, 4, 128, 128) and full.dtype == torch.float32
    assert torch.isfinite(full).all()
    # two bf16 evaluations that differ only in GEMM / conv algorithm choice decorrelate at the bf16 noise floor, which
    # guidance 7 amplifies (test_gpu_models.py measures ~3 % per step against fp32); a wrong row assignment gives O(1)
    assert _rel(torch.cat([lo, hi]), full) < 0.05
    assert _rel(torch.cat([hi, lo]), full) > 0.5
    # permuting the batch permutes the result
    perm = [2, 0, 3, 1]
    shuf = sdxl.generate_image(image_tokens=[toks[i] for i in perm], latents=noise[perm], **kw)[0]
    assert _rel(shuf, full[perm]) < 0.05


def test_config2_guidance_scale_one_equals_conditional_branch(sdxl):
    # eps_u + 1.0 * (eps_t - eps_u) = eps_t up to the bf16 roundings of the CFG arithmetic: with g = 1 the result must
    # not depend on the unconditional branch.  Checked on one full-size step against a hand-rolled conditional-only step.
    toks = _tokens(2, 3)
    noise = torch.randn(2, 4, 128, 128, generator=torch.Generator().manual_seed(5)).to(DEV)
    got = sdxl.generate_image(image_tokens=toks, latents=noise, width=1024, height=1024, num_inference_steps=1,
                              guidance_scale_for_diffusion=1.0, output_type="latent")[0]
    pe, pooled = sdxl.generate_image_embeds(toks)
    sched = sdxl.scheduler
    sched.set_timesteps(1, device=DEV)
    t = sched._host_timesteps[0]
    ids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]], dtype=pe.dtype, device=DEV).repeat(2, 1)
    eps = sdxl.unet(noise.to(torch.bfloat16), t, encoder_hidden_states=pe,
                    added_cond_kwargs={"text_embeds": pooled, "time_ids": ids}).sample
    want = sched.step(eps, t, noise).prev_sample
    assert _rel(got, want) < 0.05


def test_config3_ragged_resolution_50_step_schedule(sdxl):
    # 1216x832 -> latent 152x104: attention over N = 3952 / 988 (no tile multiple), GroupNorm rows of 15808 / 3952 /
    # 988 elements.  Three of the 50 DDIM steps on a batch of 2; independence of the batch as above.
    from pigeon_amd.pipeline import RATIO_DICT
    h, w = RATIO_DICT["2:3"]
    assert (h, w) == (1216, 832)
    toks = _tokens(2, 11)
    noise = torch.randn(2, 4, h // 8, w // 8, generator=torch.Generator().manual_seed(123)).to(DEV)
    seen = []
    kw = dict(width=w, height=h, num_inference_steps=50, guidance_scale_for_diffusion=7.0, output_type="latent")

    class Stop(Exception):
        pass

    def cb(i, t, x):
        seen.append((int(t), x.clone()))
        if i == 2:
            raise Stop

    with pytest.raises(Stop):
        sdxl.generate_image(image_tokens=toks, latents=noise, callback=cb, **kw)
    assert [t for t, _ in seen] == [981, 961, 941]          # leading spacing, offset 1, 1000 // 50 = 20
    assert all(torch.isfinite(x).all() for _, x in seen) and seen[0][1].shape == (2, 4, 152, 104)
    single = []

    def cb1(i, t, x):
        single.append(x.clone())
        if i == 2:
            raise Stop

    with pytest.raises(Stop):
        sdxl.generate_image(image_tokens=toks[1:], latents=noise[1:], callback=cb1, **kw)
    assert _rel(single[-1], seen[-1][1][1:]) < 0.06


def test_config4_multi_candidate_ordering(sdxl):
    # num_return_images = 8 (DPO preference sampling): conditioning is repeat_interleave'd (lavit_for_pigeon.py:735-736),
    # so latent row i*8 + j belongs to token sequence i.  Two sequences x 8 noise rows; feeding sequence 1 alone with
    # noise rows 8..15 must reproduce rows 8..15.  Small latent (256x256 image) keeps it quick; the stack is full size.
    toks = _tokens(2, 21)
    noise = torch.randn(16, 4, 32, 32, generator=torch.Generator().manual_seed(9)).to(DEV)
    kw = dict(width=256, height=256, num_inference_steps=2, guidance_scale_for_diffusion=7.0, output_type="latent")
    both, pe, pooled = sdxl.generate_image(image_tokens=toks, num_return_images=8, latents=noise, **kw)
    assert both.shape == (16, 4, 32, 32) and pe.shape == (2, 256, 2048) and pooled.shape == (2, 1280)
    second = sdxl.generate_image(image_tokens=toks[1:], num_return_images=8, latents=noise[8:], **kw)[0]
    assert _rel(second, both[8:]) < 0.05
    # same tokens + same noise -> same latent; same tokens + different noise -> different latent
    dup = noise.clone()
    dup[1] = dup[0]
    d2 = sdxl.generate_image(image_tokens=toks, num_return_images=8, latents=dup, **kw)[0]
    # (not bitwise: hipBLASLt's stream-K GEMMs accumulate in a tile-position-dependent order)
    assert _rel(d2[1], d2[0]) < 0.05 and _rel(d2[2], d2[0]) > 0.5


def test_config2_share_at_full_size_batch_8_1216x832(sdxl):
    # BASELINE configs[2] as bench.py --config 2 runs it: the one-GPU share of 8 images at 1216x832 on the 50-step
    # schedule (UNet batch 16, N = 3952 / 988), three steps; rows 2..3 run alone must reproduce rows 2..3 of the batch
    toks = _tokens(8, 31)
    noise = torch.randn(8, 4, 152, 104, generator=torch.Generator().manual_seed(77)).to(DEV)
    kw = dict(width=832, height=1216, num_inference_steps=50, guidance_scale_for_diffusion=7.0, output_type="latent")

    class Stop(Exception):
        pass

    def run(tk, nz):
        seen = []

        def cb(i, t, x):
            seen.append(x.clone())
            if i == 2:
                raise Stop
        with pytest.raises(Stop):
            sdxl.generate_image(image_tokens=tk, latents=nz, callback=cb, **kw)
        return seen[-1]

    full = run(toks, noise)
    assert full.shape == (8, 4, 152, 104) and torch.isfinite(full).all()
    part = run(toks[2:4], noise[2:4])
    assert _rel(part, full[2:4]) < 0.06 and _rel(part, full[4:6]) > 0.5


def test_config3_share_at_full_size_two_sequences_times_eight(sdxl):
    # BASELINE configs[3] as bench.py --config 3 runs it: 2 token sequences x 8 candidates at 1024x1024 (UNet batch 32,
    # the bench's kernel shapes), two steps, decoded: candidate rows of sequence 1 run alone reproduce rows 8..15, and
    # the images come back as 16 uint8 pictures
    toks = _tokens(2, 41)
    noise = torch.randn(16, 4, 128, 128, generator=torch.Generator().manual_seed(19)).to(DEV)
    kw = dict(width=1024, height=1024, num_inference_steps=2, guidance_scale_for_diffusion=7.0)
    both = sdxl.generate_image(image_tokens=toks, num_return_images=8, latents=noise, output_type="latent", **kw)[0]
    assert both.shape == (16, 4, 128, 128) and torch.isfinite(both).all()
    second = sdxl.generate_image(image_tokens=toks[1:], num_return_images=8, latents=noise[8:], output_type="latent", **kw)[0]
    assert _rel(second, both[8:]) < 0.05 and _rel(second, both[:8]) > 0.5
    images = sdxl.decode_latents(both[:2], "uint8")
    assert images.shape == (2, 1024, 1024, 3) and images.dtype == torch.uint8
    assert 5 < images.float().mean().item() < 250


def test_config5_high_resolution_attention_shapes():
    # 2x latent side: self-attention over N = 16384 (level 1) / 4096 (level 2); cross-attention key length swept to 1024
    from oracle import ops_ref
    from pigeon_amd import ops
    g = torch.Generator(device=DEV).manual_seed(0)
    for (B, H, Nq, Nk) in ((2, 10, 16384, 16384), (2, 20, 4096, 1024), (2, 10, 16384, 512), (1, 10, 16384, 1000)):
        C = H * 64
        q = torch.randn(B, Nq, C, device=DEV, generator=g).to(torch.bfloat16)
        k = torch.randn(B, Nk, C, device=DEV, generator=g).to(torch.bfloat16)
        v = torch.randn(B, Nk, C, device=DEV, generator=g).to(torch.bfloat16)
        o = ops.attention(q, k, torch.ones_like(v), H).float()
        assert (o - 1).abs().max().item() < 2.0 ** -7, (Nq, Nk)          # softmax rows sum to one
        o = ops.attention(q, k, v, H)
        # one (batch, head), 512 sampled query rows against the oracle
        b, h = B - 1, H - 3
        sl = slice(h * 64, (h + 1) * 64)
        rows = torch.randperm(Nq, generator=torch.Generator().manual_seed(1))[:512]
        want = ops_ref.attention_ref(q[b:b + 1, rows][..., sl].cpu(), k[b:b + 1, :, sl].cpu(), v[b:b + 1, :, sl].cpu(), 1)
        assert (o[b:b + 1, rows][..., sl].cpu().float() - want.float()).abs().max().item() < 2.0 ** -7, (Nq, Nk)


def test_config5_groupnorm_at_double_latent_side():
    # 320 channels at 256x256: groups of 655 360 elements (streaming path, 5 sweeps per lane)
    from pigeon_amd import ops
    x = (torch.randn(4, 320, 256, 256, device=DEV) * 1.3 - 0.4).to(torch.bfloat16)
    y = ops.group_norm_silu(x, 32, None, None, 1e-5, silu=False, out_dtype=None).float().view(4, 32, -1)
    assert y.mean(-1).abs().max().item() < 5e-3 and (y.var(-1, unbiased=False) - 1).abs().max().item() < 1e-2


def test_config2_bias_residual_add_full_size_equals_the_torch_pair():
    # pgn_bias_residual_add at the largest UNet activation of configs[1] ([32, 320, 128, 128], channels-last) against
    # the two torch launches it replaces, run on the same device: bit-exact; and linear in the bias at full size
    from pigeon_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(32, 320, 128, 128, device=DEV, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    r = torch.randn(32, 320, 128, 128, device=DEV, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(320, device=DEV, generator=g).bfloat16()
    want = (x + b.view(1, -1, 1, 1)) + r
    got = ops.bias_residual_add(x, b, r, inplace=False)
    assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, want)
    assert torch.equal(ops.bias_residual_add(x, torch.zeros_like(b), None, inplace=False), x)


def test_scoring_stage_full_size_properties():
    # ViT-H-14 at full size (632 M parameters, random init): features are deterministic per image, independent of the
    # batch an image is scored in, and the scores of an image against its own features are 100 / 100
    from pigeon_amd.clip_vision import VIT_H_14, CLIPScorer, CLIPVisionTransformer
    from pigeon_amd.synthetic import _materialize
    with torch.device("meta"):
        m = CLIPVisionTransformer(**VIT_H_14)
    scorer = CLIPScorer(_materialize(m, DEV, torch.bfloat16, 5), DEV, torch.bfloat16, batch_size=8)
    g = torch.Generator(device=DEV).manual_seed(9)
    images = torch.randint(0, 256, (12, 512, 512, 3), device=DEV, dtype=torch.uint8, generator=g)
    feats = scorer.encode_image(images, resize_hw=(512, 512))
    assert feats.shape == (12, 1024) and torch.isfinite(feats.float()).all()
    alone = scorer.encode_image(images[5:6], resize_hw=(512, 512))
    assert _rel(alone, feats[5:6]) < 0.03                      # batch-shape dependent GEMM tiles only
    scores = scorer.score(images, feats.float(), feats.float()[:, None, :].repeat(1, 5, 1), resize_hw=(512, 512))
    assert torch.allclose(scores, torch.full_like(scores, 100.0), atol=0.5)
    other = scorer.score(images, feats.float().roll(1, 0), None, resize_hw=(512, 512))
    # (a random-init tower maps noise images to nearly parallel features: cos ~ 0.9986, still separable from 1)
    assert (other[:, 0] < 99.95).all() and torch.isnan(other[:, 1]).all()


def test_config5_end_to_end_2048_through_generate_image(sdxl):
    # BASELINE configs[4] through the whole facade: 2048x2048 images = 256x256 latents (self-attention over N = 16384 /
    # 4096, GroupNorm rows of 65536 pixels, the VAE's 65536-token 512-wide head), batch 2, two DDIM steps, decoded to
    # uint8.  Size-independent properties: finite latents, images that use the grey range, and sample independence (a
    # sample alone reproduces its row of the batch of two up to the bf16 noise floor of another GEMM / conv algorithm).
    toks = _tokens(2, 17)
    noise = torch.randn(2, 4, 256, 256, generator=torch.Generator().manual_seed(123)).to(DEV)
    kw = dict(width=2048, height=2048, num_inference_steps=2, guidance_scale_for_diffusion=7.0)
    lat = {}
    images, pe, pooled = sdxl.generate_image(image_tokens=toks, latents=noise, output_type="uint8",
                                             callback=lambda i, t, x: lat.__setitem__(i, x.clone()), **kw)
    assert images.shape == (2, 2048, 2048, 3) and images.dtype == torch.uint8
    assert sorted(lat) == [0, 1] and lat[1].shape == (2, 4, 256, 256) and torch.isfinite(lat[1]).all()
    assert pe.shape == (2, 256, 2048) and pooled.shape == (2, 1280)
    assert images.float().std().item() > 1.0                       # not a constant image
    alone = sdxl.generate_image(image_tokens=toks[1:], latents=noise[1:], output_type="latent", **kw)[0]
    assert _rel(alone, lat[1][1:]) < 0.05 and _rel(alone, lat[1][:1]) > 0.5
    # decoding is per sample: the image of sample 1 decoded alone equals its image in the batch (same kernels, chunk 1)
    again = sdxl.decode_latents(lat[1][1:], "uint8")
    assert (again.int() - images[1:].int()).abs().float().mean().item() < 1.0
