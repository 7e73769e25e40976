"""CPU suite: pins the oracle (golden vectors produced by the reference's own modeling_decoder.py), checks the
architecture restatements against published parameter counts, and the host-side scheduler logic against the oracle
and closed-form answers."""
import os

import numpy as np
import pytest
import torch

from oracle import models_ref, ops_ref, pipeline_ref, weights
from oracle.models_ref import Policy


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "token_decoder_depth2.npz"))


@pytest.fixture(scope="module")
def decoder_sd():
    from pigeon_amd.token_decoder import HighresVQDecoder
    with torch.device("meta"):
        m = HighresVQDecoder(depth=2)
    return weights.synth_state_dict(((k, tuple(v.shape)) for k, v in m.state_dict().items()), seed=123)


def _check_against_golden(golden, prompt, pooled, atol):
    rows = torch.from_numpy(golden["prompt_rows"])
    assert np.allclose(prompt[:, rows].float().numpy(), golden["prompt_embeds_rows"], atol=atol)
    assert np.allclose(pooled.float().numpy(), golden["pooled"], atol=atol)
    assert np.allclose(prompt.double().sum(dim=(1, 2)).numpy(), golden["prompt_sum"], atol=atol * prompt[0].numel() ** 0.5 * 4)


def test_oracle_token_decoder_matches_reference_golden(golden, decoder_sd):
    tokens = list(torch.from_numpy(golden["tokens"]))
    emb, nums = ops_ref.token_gather_ref(tokens, weights.synth_codebook(), torch.float32)
    assert nums.tolist() == golden["token_nums"].tolist()
    assert np.array_equal(emb[:8].numpy(), golden["gather_rows"])
    prompt, pooled = models_ref.token_decoder_forward(Policy(torch.float32), decoder_sd, emb, nums)
    assert prompt.shape == (4, 256, 2048) and pooled.shape == (4, 1280)
    _check_against_golden(golden, prompt, pooled, atol=2e-5)


def test_lowres_vq_decoder_oracle_matches_reference_golden(golden_dir):
    # tests/golden/vq_decoder_lowres_depth2.npz: outputs of the reference's own VQDecoder (modeling_decoder.py:180-254).
    # (the PRODUCT decoders are checked against the same fixtures on the GPU: tests/test_gpu_models.py)
    from pigeon_amd.token_decoder import VQDecoder
    g = np.load(os.path.join(golden_dir, "vq_decoder_lowres_depth2.npz"))
    with torch.device("meta"):
        m = VQDecoder(depth=2)
    sd = weights.synth_like(m, seed=321)
    tokens = list(torch.from_numpy(g["tokens"]))
    emb, nums = ops_ref.token_gather_ref(tokens, weights.synth_codebook(), torch.float32)
    assert nums.tolist() == g["token_nums"].tolist()
    rows = torch.from_numpy(g["rows"])
    xrec = models_ref.token_decoder_forward(Policy(torch.float32), sd, emb, nums)
    assert xrec.shape == (4, 256, 768)
    assert np.allclose(xrec[:, rows].numpy(), g["xrec_rows"], atol=5e-5)
    assert np.allclose(xrec.double().sum(dim=(1, 2)).numpy(), g["xrec_sum"], atol=5e-5 * xrec[0].numel() ** 0.5 * 4)


def test_build_tokenizer_decoder_reads_the_reference_file_layout(tmp_path):
    # modeling_decoder.py:340-351: visual_tokenizer/{highres_tokenizer_decoder,tokenizer_decoder}.bin
    from pigeon_amd.token_decoder import HighresVQDecoder, VQDecoder, build_tokenizer_decoder
    (tmp_path / "visual_tokenizer").mkdir()
    for cls, name, mode in ((HighresVQDecoder, "highres_tokenizer_decoder.bin", "highres"),
                            (VQDecoder, "tokenizer_decoder.bin", "lowres")):
        sd = weights.synth_like(cls(depth=1), seed=3)
        torch.save(sd, tmp_path / "visual_tokenizer" / name)
        m = build_tokenizer_decoder(str(tmp_path), pixel_decoding=mode, depth=1)
        assert isinstance(m, cls) and torch.equal(m.state_dict()["in_proj.weight"], sd["in_proj.weight"])


def _nparams(m):
    return sum(p.numel() for p in m.parameters())


def test_architecture_parameter_counts_are_the_published_ones():
    from pigeon_amd.token_decoder import HighresVQDecoder
    from pigeon_amd.unet import SD15_CONFIG, SDXL_CONFIG, UNet2DConditionModel
    from pigeon_amd.vae import AutoencoderKL
    with torch.device("meta"):
        assert _nparams(UNet2DConditionModel(**SDXL_CONFIG)) == 2_567_463_684      # SDXL-base UNet
        assert _nparams(UNet2DConditionModel(**SD15_CONFIG)) == 859_520_964        # SD-1.5 UNet
        assert _nparams(AutoencoderKL().decoder) == 49_490_179                     # SD/SDXL VAE decoder
        n = _nparams(HighresVQDecoder(depth=12))
    assert abs(n - 414.0e6) < 0.06e6   # 414.0 M measured on the imported reference module


def test_unet_state_dict_uses_diffusers_names():
    from pigeon_amd.unet import SDXL_CONFIG, UNet2DConditionModel
    with torch.device("meta"):
        keys = set(UNet2DConditionModel(**SDXL_CONFIG).state_dict())
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "add_embedding.linear_2.bias",
              "down_blocks.0.resnets.1.norm2.weight", "down_blocks.0.downsamplers.0.conv.weight",
              "down_blocks.1.attentions.0.transformer_blocks.1.attn1.to_q.weight",
              "down_blocks.2.attentions.1.transformer_blocks.9.attn2.to_out.0.bias",
              "down_blocks.2.attentions.1.transformer_blocks.9.ff.net.0.proj.weight",
              "down_blocks.2.attentions.1.transformer_blocks.9.ff.net.2.weight",
              "mid_block.attentions.0.proj_in.weight", "mid_block.resnets.1.time_emb_proj.weight",
              "up_blocks.0.resnets.2.conv_shortcut.weight", "up_blocks.1.upsamplers.0.conv.bias",
              "up_blocks.2.resnets.0.norm1.weight", "conv_norm_out.weight", "conv_out.bias"):
        assert k in keys, k
    assert "down_blocks.2.downsamplers.0.conv.weight" not in keys
    assert not any("attentions" in k for k in keys if k.startswith(("down_blocks.0.", "up_blocks.2.")))


SCHED = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
             steps_offset=1, clip_sample=False, set_alpha_to_one=False)


@pytest.mark.parametrize("spacing", ["leading", "linspace", "trailing"])
@pytest.mark.parametrize("n", [25, 50, 7])
def test_ddim_schedule_matches_oracle_and_known_answers(spacing, n):
    from pigeon_amd.schedulers import DDIMScheduler
    s = DDIMScheduler(timestep_spacing=spacing, **SCHED)
    r = pipeline_ref.DDIMRef(timestep_spacing=spacing, **SCHED)
    s.set_timesteps(n)
    r.set_timesteps(n)
    assert s.timesteps.tolist() == r.timesteps.tolist() == s._host_timesteps
    assert torch.equal(s.alphas_cumprod, r.alphas_cumprod)
    if spacing == "leading" and n == 25:
        assert s._host_timesteps == list(range(961, 0, -40))          # 1 + 40 k
    if spacing == "trailing" and n == 25:
        assert s._host_timesteps[0] == 999 and s._host_timesteps[-1] == 39
    assert abs(float(s.alphas_cumprod[0]) - 0.99915) < 1e-6 and abs(float(s.alphas_cumprod[-1]) - 0.0046602) < 1e-5
    # per-step coefficients are the fp32 values the oracle's step derives
    for t in s._host_timesteps:
        sa_t, sb_t, sa_p, dirc = s._coef[t]
        prev_t = t - 1000 // n
        a_t = r.alphas_cumprod[t]
        a_p = r.alphas_cumprod[prev_t] if prev_t >= 0 else r.final_alpha_cumprod
        assert (sa_t, sb_t, sa_p, dirc) == (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5),
                                             float((1 - a_p - 0.0 ** 2) ** 0.5))
    assert s.init_noise_sigma == 1.0 and s.scale_divisor(s._host_timesteps[0]) == 1.0


def test_ddim_errors_mirror_diffusers():
    from pigeon_amd.schedulers import DDIMScheduler
    s = DDIMScheduler(**SCHED)
    with pytest.raises(ValueError):
        s.set_timesteps(1001)
    with pytest.raises(ValueError):
        s.step(torch.zeros(1), 1, torch.zeros(1))   # set_timesteps not called
    with pytest.raises(ValueError):
        DDIMScheduler(timestep_spacing="bogus", **SCHED).set_timesteps(10)


def test_euler_schedule_matches_oracle():
    from pigeon_amd.schedulers import EulerDiscreteScheduler
    cfg = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="leading",
               steps_offset=1)
    s, r = EulerDiscreteScheduler(**cfg), pipeline_ref.EulerRef(**cfg)
    assert float(s.init_noise_sigma) == float(r.init_noise_sigma)
    s.set_timesteps(25)
    r.set_timesteps(25)
    assert torch.equal(s.sigmas, r.sigmas) and s.timesteps.tolist() == r.timesteps.tolist()
    assert s.scale_divisor(s.timesteps[0]) == float((r.sigmas[0] ** 2 + 1) ** 0.5)
    assert abs(float(s.sigmas[0]) - 13.0) < 1.7 and float(s.sigmas[-1]) == 0.0   # SDXL sigma_max ~ 14.6 at t=999


def test_product_has_no_cpu_fallback():
    from pigeon_amd import ops
    with pytest.raises(RuntimeError):
        ops.group_norm_silu(torch.zeros(1, 32, 4, 4), 32, None, None)
    with pytest.raises(RuntimeError):
        ops.attention(torch.zeros(1, 4, 64), torch.zeros(1, 4, 64), torch.zeros(1, 4, 64), 1)
    with pytest.raises(RuntimeError):
        ops.cfg_ddim_step(torch.zeros(2, 4), torch.zeros(1, 4), 7.0, 1, 1, 1, 1)
    # ... and neither have the modules built on them: token decoder, CLIP tower, VAE decoder, UNet on CPU tensors raise
    # instead of evaluating with torch ops
    from pigeon_amd.clip_vision import CLIPVisionTransformer
    from pigeon_amd.token_decoder import HighresVQDecoder, VQDecoder
    from pigeon_amd.unet import UNet2DConditionModel
    from pigeon_amd.vae import AutoencoderKL
    with torch.no_grad():
        for cls in (HighresVQDecoder, VQDecoder):
            m = cls(depth=1).eval()
            with pytest.raises(RuntimeError, match="no CPU fallback"):
                m(torch.zeros(6, 32), torch.tensor([2, 4]))
            with pytest.raises(RuntimeError, match="no CPU fallback"):
                m.forward_padded(torch.zeros(2, 256, 32), torch.ones(2, 256))
        vit = CLIPVisionTransformer(image_size=28, patch_size=14, width=64, layers=1, heads=2, output_dim=16).eval()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            vit(torch.zeros(1, 3, 28, 28))
        vae = AutoencoderKL(block_out_channels=(32, 64), layers_per_block=1).eval()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            vae.decode(torch.zeros(1, 4, 8, 8))
        unet = UNet2DConditionModel(block_out_channels=(32, 64), attention_head_dim=(1, 1), layers_per_block=1,
                                    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
                                    up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"),
                                    transformer_layers_per_block=1, cross_attention_dim=64).eval()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            unet(torch.zeros(1, 4, 8, 8), 10, encoder_hidden_states=torch.zeros(1, 4, 64),
                 added_cond_kwargs={"text_embeds": torch.zeros(1, 1280), "time_ids": torch.zeros(1, 6)})


def test_threshold_sample_known_answers():
    # dynamic thresholding by hand: |x| = 0..9 -> 0.9-quantile = 8.1 (linear interpolation), clamped to max 4 -> s = 4;
    # a sample whose quantile is below 1 keeps s = 1 (plain clip to [-1, 1])
    from oracle.pipeline_ref import threshold_sample_ref
    x = torch.tensor([[0., -1, 2, -3, 4, -5, 6, -7, 8, -9], [0.05 * i for i in range(10)]]).reshape(2, 10, 1, 1)
    y = threshold_sample_ref(x, 0.9, 4.0).reshape(2, 10)
    assert torch.allclose(y[0], torch.tensor([0., -0.25, 0.5, -0.75, 1, -1, 1, -1, 1, -1]))
    assert torch.allclose(y[1], x.reshape(2, 10)[1])
    y = threshold_sample_ref(x, 0.9, 100.0).reshape(2, 10)
    assert torch.allclose(y[0], (x.reshape(2, 10)[0] / 8.1).clamp(-1, 1))
    # and the torch.quantile form diffusers uses gives the same numbers
    q = torch.quantile(x.reshape(2, 10).abs(), 0.9, dim=1)
    assert torch.allclose(q, torch.tensor([8.1, 0.405]))
