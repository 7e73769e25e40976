"""GPU parity tests of the four HIP kernels, called through the C ABI (pigeon_amd.ops -> ctypes -> libpigeon_hip).

Bar (SURVEY.md §8 / BASELINE.json north_star): bit-exact for the integer/byte work (gather) and for the step
kernels in reference-rounding mode; stated tolerances for GroupNorm+SiLU and attention (fp32 math on bf16 data,
one bf16 rounding of the result).
"""
import math

import pytest
import torch

from oracle import ops_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pigeon_amd import ops as _ops
    return _ops


DEV = "cuda"


def _codebook(seed=0, n=16384, d=32):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(n, d, generator=g)
    return torch.nn.functional.normalize(w, p=2, dim=-1)  # modeling_visual_tokenzier.py:59-61


def _token_cases():
    g = torch.Generator().manual_seed(123)
    def valid(n):
        return torch.randint(32002, 32002 + 16384, (n,), generator=g)
    cases = []
    # what llama.generate returns: [valid ids] + eos(32001) + pad(2) up to the batch max (inference.py:268-282)
    a = torch.cat([valid(90), torch.tensor([32001]), torch.full((40,), 2)])
    b = torch.cat([torch.tensor([32000]), valid(255), torch.tensor([32001])])
    c = torch.cat([valid(300)])                                  # > 256 valid ids -> truncated
    d = torch.tensor([2, 2, 32000, 32001, 5, 31999])             # nothing survives the filter
    e = torch.cat([valid(20), torch.tensor([2, 32001]), valid(7)])  # specials in the middle
    f = valid(256)
    cases.append([a, b, c, d, e, f])
    cases.append([valid(1)])
    cases.append([valid(n) for n in (64, 65, 63, 128, 1, 256, 257, 300, 299)] + [d])
    return cases


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
def test_token_gather_bit_exact(ops, dtype):
    cb = _codebook()
    for toks in _token_cases():
        want, want_nums = ops_ref.token_gather_ref(toks, cb, dtype)
        got = ops.token_gather([t.to(DEV) for t in toks], cb.to(DEV), out_dtype=dtype, check=True)
        nums = got["token_nums"].cpu()
        assert torch.equal(nums, want_nums)
        total = int(want_nums.sum())
        assert torch.equal(got["packed"][:total].cpu().view(torch.uint8), want.view(torch.uint8))
        pad, mask = ops_ref.pad_and_mask_ref(want, want_nums, pad_len=256)
        assert torch.equal(got["padded"].cpu().view(torch.uint8), pad.view(torch.uint8))
        assert torch.equal(got["mask"].cpu().view(torch.uint8), mask.view(torch.uint8))


def test_token_gather_rectangular_batch_and_bad_id(ops):
    cb = _codebook(1)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(32002, 32002 + 16384, (16, 301), generator=g)
    ids[:, 200:] = 2
    ids[3, 150] = 32001
    want, nums = ops_ref.token_gather_ref(list(ids), cb)
    got = ops.token_gather(ids.to(DEV), cb.to(DEV), check=True)
    assert torch.equal(got["token_nums"].cpu(), nums)
    assert torch.equal(got["packed"][: int(nums.sum())].cpu().view(torch.int16), want.view(torch.int16))
    bad = ids.clone()
    bad[0, 0] = 32002 + 16384  # F.embedding would raise IndexError
    with pytest.raises(IndexError):
        ops.token_gather(bad.to(DEV), cb.to(DEV), check=True)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_token_embed_matches_gather_then_in_proj_mask_pos_embed(ops, dtype):
    # pgn_token_embed = pgn_token_gather carried through modeling_decoder.py:310-319 (in_proj, x mask, + pos_embed)
    # and :100-101 (key bias).  Checked against the exact statement of those lines on the bit-exact gather's output:
    # fp64 dot of the 16-bit rows and weights + bias -> rounded to the 16-bit type (the GEMM's output), x mask,
    # + pos_embed, rounded.  Only the fp32 accumulation order can differ: the GEMM output may land on the other side of
    # a rounding tie (1 ulp of it) on a few elements, nothing else; padding rows are pos_embed exactly.
    cb = _codebook(2)
    g = torch.Generator().manual_seed(17)
    E = 1408
    w = (torch.randn(E, 32, generator=g) * 0.2).to(dtype)
    b = (torch.randn(E, generator=g) * 0.1).to(dtype)
    pos = (torch.randn(1, 256, E, generator=g) * 0.5).to(dtype)
    # LDS is not cleared between kernels: leave NaN bit patterns in every CU's LDS (an attention launch over NaN
    # inputs stages them as K / V tiles) so that a kernel reading LDS it has not written shows up as NaN output
    # (regression: the dot product ran over 64 columns of which only `dim` = 32 had been filled)
    nan = torch.full((64, 512, 640), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.attention(nan, nan, nan, 10)
    for toks in _token_cases():
        dev_toks = [t.to(DEV) for t in toks]
        ref = ops.token_gather(dev_toks, cb.to(DEV), out_dtype=dtype, packed=False)
        got = ops.token_embed(dev_toks, cb.to(DEV), w.to(DEV), b.to(DEV), pos.to(DEV), check=True)
        assert torch.equal(got["token_nums"], ref["token_nums"])
        mask = ref["mask"].cpu().double()
        assert torch.equal(got["key_bias"].cpu().double(), (1.0 - mask) * -10000.0)
        y = (ref["padded"].cpu().double() @ w.double().T + b.double()).to(dtype)
        want = ((y.double() * mask[..., None]).to(dtype).double() + pos.double()).to(dtype)
        x = got["x"].cpu()
        assert x.shape == want.shape and x.dtype == dtype and bool(torch.isfinite(x.float()).all())
        pad_rows = mask == 0
        assert torch.equal(x[pad_rows], pos.expand_as(x)[pad_rows])
        # one ulp of the GEMM output y (the value whose rounding can flip) ...
        mant = 7 if dtype == torch.bfloat16 else 10
        ulp = lambda t: torch.exp2(torch.floor(torch.log2(t.abs().double().clamp_min(2.0 ** -24))) - mant)
        err = (x.double() - want.double()).abs()
        bound = ulp(y) * mask[..., None] + ulp(want) + 1e-12                 # ... and a flipped rounding of the sum
        frac = float((err != 0).double().mean())
        print(f"token_embed {dtype}: {frac:.2e} of the elements differ, worst err/bound {float((err / bound).max()):.2f}")
        assert bool((err <= bound).all()), float((err / bound).max())
        assert frac < 2e-3
    with pytest.raises(TypeError):
        ops.token_embed(dev_toks, cb.to(DEV), w.float().to(DEV), None, pos.float().to(DEV))
    with pytest.raises(NotImplementedError):
        ops.token_embed(dev_toks, torch.randn(64, 128, device=DEV), torch.randn(E, 128, device=DEV).to(dtype), None,
                        pos.to(DEV))
    bad = [torch.tensor([32002 + 16384])]
    with pytest.raises(IndexError):
        ops.token_embed([t.to(DEV) for t in bad], cb.to(DEV), w.to(DEV), b.to(DEV), pos.to(DEV), check=True)


def _step_inputs(shape, edt, seed=0):
    g = torch.Generator().manual_seed(seed)
    noise_pred = torch.randn((2 * shape[0],) + shape[1:], generator=g).to(edt)
    sample = torch.randn(shape, generator=g) * 3.0
    return noise_pred, sample


@pytest.mark.parametrize("edt", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("pred", ["epsilon", "sample", "v_prediction"])
def test_cfg_ddim_step_bit_exact(ops, edt, pred):
    # [B,4,h,w] incl. the BASELINE config-2 size (16x4x128x128) and a size with a scalar tail
    for shape, g in (((16, 4, 128, 128), 7.0), ((3, 4, 19, 13), 7.1), ((1, 4, 152, 104), 5.0)):
        noise_pred, sample = _step_inputs(shape, edt)
        a_t, a_p = torch.tensor(0.2174), torch.tensor(0.3021)
        for clip, ucm in ((None, False), (1.0, False), (1.0, True)):
            want_prev, want_x0 = ops_ref.ddim_step_ref(ops_ref.cfg_combine_ref(noise_pred, g), sample, a_t, a_p, pred,
                                                        clip, ucm)
            got_prev, got_x0 = ops.cfg_ddim_step(noise_pred.to(DEV), sample.to(DEV), g, float(a_t ** 0.5),
                                                 float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5),
                                                 prediction_type=pred, clip_sample_range=clip,
                                                 use_clipped_model_output=ucm, return_pred_x0=True)
            assert got_prev.dtype == want_prev.dtype == torch.float32
            assert torch.equal(got_prev.cpu(), want_prev), (shape, clip, ucm)
            assert torch.equal(got_x0.cpu(), want_x0.float()), (shape, clip, ucm)


def test_cfg_ddim_step_fused_fp32_is_close_and_no_guidance_path(ops):
    noise_pred, sample = _step_inputs((4, 4, 64, 64), torch.bfloat16)
    a_t, a_p = torch.tensor(0.5), torch.tensor(0.7)
    args = (7.0, float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5))
    exact, _ = ops.cfg_ddim_step(noise_pred.to(DEV), sample.to(DEV), *args)
    fused, _ = ops.cfg_ddim_step(noise_pred.to(DEV), sample.to(DEV), *args, fused_fp32=True)
    # fused mode skips three bf16 roundings of O(10) magnitude values: tolerance 2^-7 relative of |eps_cfg|
    assert (exact - fused).abs().max().item() < 0.25
    text = noise_pred[4:]
    want, _ = ops_ref.ddim_step_ref(text, sample, a_t, a_p)
    got, _ = ops.cfg_ddim_step(None, sample.to(DEV), *args, eps_text=text.to(DEV))
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction", "sample"])
def test_cfg_euler_step_bit_exact(ops, pred):
    noise_pred, sample = _step_inputs((2, 4, 64, 64), torch.bfloat16, seed=3)
    sigma, sigma_next = torch.tensor(7.3125), torch.tensor(5.0411)
    want, want_x0 = ops_ref.euler_step_ref(ops_ref.cfg_combine_ref(noise_pred, 7.0), sample, sigma, sigma_next, pred)
    got, got_x0 = ops.cfg_euler_step(noise_pred.to(DEV), sample.to(DEV), 7.0, float(sigma), float(sigma_next), pred,
                                     return_pred_x0=True)
    assert got.dtype == torch.bfloat16  # diffusers casts prev_sample back to model_output.dtype
    assert torch.equal(got.cpu(), want)
    # second step: bf16 latents in
    want2, _ = ops_ref.euler_step_ref(ops_ref.cfg_combine_ref(noise_pred, 7.0), want, sigma_next, torch.tensor(3.3), pred)
    got2, _ = ops.cfg_euler_step(noise_pred.to(DEV), got, 7.0, float(sigma_next), 3.3, pred)
    assert torch.equal(got2.cpu(), want2)
    # fp32-latent variant keeps full precision
    got32, _ = ops.cfg_euler_step(noise_pred.to(DEV), sample.to(DEV), 7.0, float(sigma), float(sigma_next), pred,
                                  out_dtype=torch.float32)
    want32 = ops_ref.euler_step_ref(ops_ref.cfg_combine_ref(noise_pred, 7.0).float(), sample, sigma, sigma_next, pred)[0]
    assert got32.dtype == torch.float32
    assert (got32.cpu() - want32).abs().max() < 0.3  # bf16 rounding of sigma*eps, |.| up to ~40


def test_cfg_expand_scale(ops):
    x = torch.randn(4, 4, 32, 32)
    for div in (1.0, float((torch.tensor(7.3) ** 2 + 1) ** 0.5)):
        got = ops.cfg_expand_scale(x.to(DEV), torch.bfloat16, div)
        want = torch.cat([x / div] * 2).to(torch.bfloat16)
        assert torch.equal(got.cpu(), want)


# (N, C, H, W, groups, eps, silu, dtype, affine)
GN_CASES = [
    (2, 320, 128, 128, 32, 1e-5, True, torch.bfloat16, True),    # streaming path, 10 ch x 16384
    (2, 640, 64, 64, 32, 1e-5, True, torch.bfloat16, True),      # resident R=16
    (2, 1280, 32, 32, 32, 1e-5, True, torch.bfloat16, True),     # resident R=8
    (2, 2560, 32, 32, 32, 1e-5, True, torch.bfloat16, True),     # resident R=16 (80 ch x 1024)
    (1, 960, 128, 128, 32, 1e-5, True, torch.bfloat16, True),    # streaming, 30 ch x 16384
    (2, 640, 64, 64, 32, 1e-6, False, torch.bfloat16, True),     # Transformer2DModel.norm (no SiLU)
    (2, 640, 76, 52, 32, 1e-5, True, torch.bfloat16, True),      # ragged 1216x832 level 1 (HW = 3952)
    (2, 1280, 38, 26, 32, 1e-5, True, torch.bfloat16, True),     # HW = 988, vectors straddle channels
    (2, 1920, 38, 26, 32, 1e-5, True, torch.bfloat16, True),
    (3, 64, 16, 16, 32, 1e-5, True, torch.float32, True),        # fp32 config
    (2, 128, 8, 8, 32, 1e-5, True, torch.float16, True),
    (2, 64, 5, 3, 32, 1e-5, True, torch.bfloat16, False),        # scalar path, no affine
    (2, 64, 1, 1, 32, 1e-5, False, torch.float32, True),         # two elements per group
    (1, 128, 512, 512, 32, 1e-6, True, torch.bfloat16, True),    # split path (workspace): VAE decoder, 32 pairs x 1 Mi
    (2, 64, 256, 256, 32, 1e-6, True, torch.bfloat16, True),     # split path, 64 pairs x 128 Ki
    (1, 256, 131, 127, 32, 1e-6, False, torch.float16, True),    # split path, vectors straddle channels, ragged slice
]


@pytest.mark.parametrize("case", GN_CASES, ids=[f"{c[1]}x{c[2]}x{c[3]}-{str(c[7]).split('.')[-1]}" for c in GN_CASES])
def test_groupnorm_silu_matches_oracle(ops, case):
    N, C, H, W, G, eps, silu, dtype, affine = case
    g = torch.Generator().manual_seed(C + H)
    # per-channel offsets and scales so mean != 0 and groups differ
    x = (torch.randn(N, C, H, W, generator=g) * (0.5 + torch.rand(1, C, 1, 1, generator=g) * 2)
         + torch.randn(1, C, 1, 1, generator=g)).to(dtype)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype) if affine else None
    b = (0.2 * torch.randn(C, generator=g)).to(dtype) if affine else None
    want = ops_ref.group_norm_silu_ref(x, G, w, b, eps, silu)
    got = ops.group_norm_silu(x.to(DEV), G, None if w is None else w.to(DEV), None if b is None else b.to(DEV), eps,
                              silu).cpu()
    assert got.dtype == dtype and got.shape == x.shape
    err = (got.float() - want.float()).abs()
    # tolerance: fp32 arithmetic both sides, one rounding to the io dtype -> at most 1 ulp of the io dtype
    # (fp32 io: 1e-5 relative + 1e-5 absolute covers v_exp_f32 / v_rcp_f32 and summation-order differences)
    ulp = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10, torch.float32: 1e-5}[dtype]
    bound = ulp * want.float().abs() + (1e-5 if dtype == torch.float32 else 4 * ulp * 2.0 ** -7)
    assert bool((err <= bound + 1e-6).all()), f"max err {err.max().item()} at {want.flatten()[err.argmax()].item()}"
    if dtype != torch.float32:
        assert (got != want).float().mean().item() < 0.02  # only rare 1-ulp rounding flips


@pytest.mark.parametrize("shape", [(4, 320, 128, 128), (2, 640, 64, 64), (2, 1280, 38, 26), (2, 64, 5, 3)],
                         ids=["stream", "resident", "straddle", "scalar"])
def test_groupnorm_fused_channel_add_equals_add_then_norm(ops, shape):
    # ResnetBlock2D: norm2(h + temb[:, :, None, None]) — the kernel adds, rounds to bf16 and normalises in one pass;
    # bit-identical to the kernel applied to the separately added (bf16) tensor, and within tolerance of the oracle
    N, C, H, W = shape
    g = torch.Generator().manual_seed(C)
    x = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16)
    t = (torch.randn(N, C, generator=g) * 0.5).to(torch.bfloat16)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(torch.bfloat16)
    b = (0.2 * torch.randn(C, generator=g)).to(torch.bfloat16)
    summed = x + t[:, :, None, None]                       # torch bf16 add (CPU): fp32 add, one rounding
    want = ops_ref.group_norm_silu_ref(summed, 32, w, b, 1e-5, True)
    fused = ops.group_norm_silu(x.to(DEV), 32, w.to(DEV), b.to(DEV), 1e-5, True, add_nc=t.to(DEV))
    two_step = ops.group_norm_silu(summed.to(DEV), 32, w.to(DEV), b.to(DEV), 1e-5, True)
    assert torch.equal(fused, two_step)
    err = (fused.cpu().float() - want.float()).abs()
    assert bool((err <= 2.0 ** -7 * want.float().abs() + 2.0 ** -12).all())


def test_groupnorm_split_path_is_selected_and_deterministic(ops):
    from pigeon_amd import _lib
    lib = _lib.load()
    assert lib.pgn_groupnorm_workspace_bytes(1, 128, 1024 * 1024, 32) == 32 * 32 * 8     # VAE out block: 32 slices
    assert lib.pgn_groupnorm_workspace_bytes(32, 320, 16384, 32) == 0                    # UNet at B' = 32: one launch
    x = (torch.randn(1, 128, 512, 512, device=DEV) * 2 + 1).to(torch.bfloat16)
    w, b = torch.randn(128, device=DEV).bfloat16(), torch.randn(128, device=DEV).bfloat16()
    y1 = ops.group_norm_silu(x, 32, w, b, 1e-6, True)
    y2 = ops.group_norm_silu(x, 32, w, b, 1e-6, True)
    assert torch.equal(y1, y2)   # fixed-order reduction of the partials: run-to-run bit-identical
    # same values as the one-launch kernel up to the io rounding (statistics summed in a different order)
    rc_y = torch.empty_like(x)
    rc = lib.pgn_groupnorm_silu(x.data_ptr(), 1, rc_y.data_ptr(), 1, w.data_ptr(), b.data_ptr(), 1, 1, 128, 512 * 512,
                                32, 1e-6, 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert (rc_y != y1).float().mean().item() < 0.01
    assert (rc_y.float() - y1.float()).abs().max().item() <= 2.0 ** -6 * y1.float().abs().max().item()


def test_groupnorm_full_size_statistics_property(ops):
    # BASELINE config-2 size (B' = 32): too big for the CPU oracle in seconds -> size-independent property:
    # without affine/activation every (sample, group) of the output has mean 0 and variance 1.
    N, C, HW, G = 32, 320, 16384, 32
    x = (torch.randn(N, C, HW, device=DEV) * 1.7 + 0.3).to(torch.bfloat16)
    y = ops.group_norm_silu(x, G, None, None, 1e-5, silu=False, out_dtype=torch.float32)
    yg = y.view(N, G, -1)
    assert yg.mean(-1).abs().max().item() < 2e-3
    assert (yg.var(-1, unbiased=False) - 1).abs().max().item() < 2e-3


# (B, heads, Nq, Nk, dtype, bias)
ATTN_CASES = [
    (2, 10, 1024, 1024, torch.bfloat16, False),   # SDXL level-2 self-attention shape (heads reduced)
    (1, 10, 4096, 4096, torch.bfloat16, False),   # SDXL level-1 self-attention
    (2, 20, 1024, 256, torch.bfloat16, False),    # cross-attention to the 256 decoder tokens
    (2, 5, 988, 988, torch.bfloat16, False),      # ragged: 1216x832 level 2
    (1, 2, 3952, 3952, torch.bfloat16, False),    # ragged: 1216x832 level 1
    (2, 4, 300, 77, torch.bfloat16, True),        # key padding bias, Nk tail
    (3, 2, 1, 1, torch.bfloat16, False),
    (2, 3, 257, 65, torch.float16, False),
    (2, 8, 1024, 256, torch.bfloat16, True),      # lowres-style encoder_attention_mask
    # the 64-rows-per-wave LDS-DMA kernel (head dim 64, no bias, Nk >= 1024, Nq >= 512 and at least 1024 of its
    # 256-row workgroups): ragged on both sides, fp16
    (16, 16, 1000, 1100, torch.bfloat16, False),
    (8, 32, 1024, 1536, torch.float16, False),
    (22, 16, 520, 1025, torch.bfloat16, False),   # one key in the last tile, 8 rows in the last workgroup
    (1, 2, 1000, 1100, torch.bfloat16, False),    # the same shapes below the workgroup threshold: the 32-row kernel
    (2, 2, 520, 1025, torch.bfloat16, False),
]


def _attn_err_over_bar(got, q, k, v, H, bias=None):
    """worst |got - oracle| / (2^-7 max(1, max|oracle|)) with the fp32 oracle evaluated per batch (bounded host memory)"""
    worst = 0.0
    for b in range(q.shape[0]):
        want = ops_ref.attention_ref(q[b:b + 1], k[b:b + 1], v[b:b + 1], H, None,
                                     None if bias is None else bias[b:b + 1]).float()
        worst = max(worst, (got[b:b + 1] - want).abs().max().item() / (2.0 ** -7 * max(1.0, want.abs().max().item())))
    return worst


def _attn_inputs(B, H, Nq, Nk, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    C = H * 64
    q = (torch.randn(B, Nq, C, generator=g) * 1.5).to(dtype)
    k = (torch.randn(B, Nk, C, generator=g) * 1.5).to(dtype)
    v = torch.randn(B, Nk, C, generator=g).to(dtype)
    return q, k, v


@pytest.mark.parametrize("case", ATTN_CASES, ids=[f"B{c[0]}H{c[1]}q{c[2]}k{c[3]}{'b' if c[5] else ''}" for c in ATTN_CASES])
def test_attention_matches_oracle(ops, case):
    B, H, Nq, Nk, dtype, has_bias = case
    q, k, v = _attn_inputs(B, H, Nq, Nk, dtype, Nq + Nk)
    bias = None
    if has_bias:
        keep = torch.randint(1, Nk + 1, (B,), generator=torch.Generator().manual_seed(1))
        mask = (torch.arange(Nk)[None, :] < keep[:, None]).float()
        bias = (1.0 - mask) * -10000.0  # modeling_decoder.py:100-101 / diffusers encoder_attention_mask
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), H, None, None if bias is None else bias.to(DEV)).cpu().float()
    # tolerance: P is rounded to bf16 before P.V (as in xformers' kernel) and the output once more:
    # 2^-8 relative on O(1) values -> bar 2^-7 max(1, max|want|)
    err = _attn_err_over_bar(got, q, k, v, H, bias)
    assert err < 1.0, f"max abs err / bar {err}"


def test_attention_strided_inputs_and_rescale_branch(ops):
    # q/k/v as slices of one fused qkv projection output (non-contiguous rows), plus a late score spike that
    # forces the online-softmax rescale in a non-first K tile (cdna guide rule 26)
    B, H, N = 2, 3, 640
    C = H * 64
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B, N, 3 * C, generator=g).to(torch.bfloat16)
    qkv[:, 500, C:2 * C] = qkv[:, 17, 0:C] * 4.0   # key 500 aligned with query 17 -> max jumps in tile 7
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    want = ops_ref.attention_ref(q, k, v, H).float()
    dq = qkv.to(DEV)
    got = ops.attention(dq[..., :C], dq[..., C:2 * C], dq[..., 2 * C:], H).cpu().float()
    assert (got - want).abs().max().item() < 2.0 ** -7 * max(1.0, want.abs().max().item())


def test_attention_rescale_branch_of_the_64_row_kernel(ops):
    # same late-spike construction on a shape the 64-rows-per-wave kernel takes (Nk >= 1024): the deferred softmax
    # reference (raised only when a row maximum outgrows it by 2^8) must be raised in a late tile for SOME rows of a
    # block (key 1500 aligned with queries 17 and 300) and left alone for moderate growth (key 700, query 40)
    B, H, N = 10, 15, 1600                  # 1050 workgroups of 256 rows: the 64-row kernel's side of the heuristic
    C = H * 64
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B, N, 3 * C, generator=g).to(torch.bfloat16)
    qkv[:, 1500, C:2 * C] = (qkv[:, 17, 0:C].float() * 2.5 + qkv[:, 300, 0:C].float() * 2.5).to(torch.bfloat16)
    qkv[:, 700, C:2 * C] = qkv[:, 40, 0:C] * 0.5
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    dq = qkv.to(DEV)
    got = ops.attention(dq[..., :C], dq[..., C:2 * C], dq[..., 2 * C:], H).cpu().float()
    assert _attn_err_over_bar(got, q, k, v, H) < 1.0
    # the spike really dominates rows 17 and 300 (their output is close to V row 1500), i.e. the branch was exercised
    want0 = ops_ref.attention_ref(q[:1], k[:1], v[:1], H).float()
    assert (want0[:, 17] - v[:1, 1500].float()).abs().max().item() < 0.25


@pytest.mark.parametrize("Nk", [1030, 1100, 1160, 1217], ids=["17_tiles_rem6", "18_tiles_rem12", "19_tiles_rem8", "20_tiles_rem1"])
def test_attention_ring_kernel_ragged_last_tile(ops, Nk):
    """Ragged key counts on the ring kernel: the last tile (in ring buffer 1, 2, 0 and 1 for these tile counts: every
    position the masked tile body is instantiated at) re-reads its last real row for the missing ones and masks their
    scores to -inf.  The very last key carries a spike for one query (the reference is raised inside the masked tile),
    the padding rows of V are never reached.  Every output row against the fp32 oracle."""
    B, H, Nq = 8, 16, 2048                        # 1024 workgroups of 256 rows: the 64-row class
    C = H * 64
    g = torch.Generator().manual_seed(Nk)
    q = torch.randn(B, Nq, C, generator=g)
    k = torch.randn(B, Nk, C, generator=g)
    v = torch.randn(B, Nk, C, generator=g)
    k[:, Nk - 1] = q[:, 33] * 3.0
    q, k, v = q.bfloat16(), k.bfloat16(), v.bfloat16()
    # k / v as the tail of a larger buffer: what follows the last real row in memory is NaN, not zeros
    kbuf = torch.full((B, Nk + 70, C), float("nan"), dtype=torch.bfloat16)
    vbuf = torch.full((B, Nk + 70, C), float("nan"), dtype=torch.bfloat16)
    kbuf[:, :Nk] = k
    vbuf[:, :Nk] = v
    got = ops.attention(q.to(DEV), kbuf.to(DEV)[:, :Nk], vbuf.to(DEV)[:, :Nk], H)
    assert torch.isfinite(got).all()
    assert _attn_err_over_bar(got.cpu().float(), q, k, v, H) < 1.0


def test_attention_64_row_class_random_shapes(ops):
    """Twenty seeded random shapes inside the 64-row class (head dim 64, no bias, Nq >= 512, Nk >= 256, enough
    workgroups): query counts that are not multiples of 256, every remainder of the key count modulo 64 and of the tile
    count modulo 3, strided q / k / v views of a fused projection.  Every output row against the fp32 oracle."""
    rng = torch.Generator().manual_seed(20251111)
    for case in range(20):
        Nq = int(torch.randint(512, 1400, (1,), generator=rng))
        Nk = int(torch.randint(256, 1400, (1,), generator=rng))
        if case % 4 == 0:
            Nk = Nq                                   # self-attention through one fused qkv buffer
        H = int(torch.randint(2, 6, (1,), generator=rng)) * 4
        B = (1024 + H * ((Nq + 255) // 256) - 1) // (H * ((Nq + 255) // 256))   # >= 1024 workgroups of 256 rows
        C = H * 64
        if Nk == Nq:
            qkv = (torch.randn(B, Nq, 3 * C, generator=rng) * 1.2).bfloat16()
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            dq = qkv.to(DEV)
            got = ops.attention(dq[..., :C], dq[..., C:2 * C], dq[..., 2 * C:], H)
        else:
            q = (torch.randn(B, Nq, C, generator=rng) * 1.2).bfloat16()
            kv = (torch.randn(B, Nk, 2 * C, generator=rng) * 1.2).bfloat16()
            k, v = kv[..., :C], kv[..., C:]
            dkv = kv.to(DEV)
            got = ops.attention(q.to(DEV), dkv[..., :C], dkv[..., C:], H)
        err = _attn_err_over_bar(got.cpu().float(), q, k, v, H)
        assert err < 1.0, (case, B, H, Nq, Nk, err)


@pytest.mark.parametrize("pattern", ["steady_growth", "first_tile_dominates", "huge_scores", "tiny_scores"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_attention_ring_kernel_sum_check_paths(ops, pattern, dtype):
    """The ring kernel takes no row maximum on its common path: a tile's probabilities are taken against the current
    reference and the lane's sum of them is the overflow check (attention.hip).  Inputs that drive that check:
    scores that grow from tile to tile (the reference has to be raised again and again, in the middle of a row block),
    a first tile that dominates everything after it (the reference never moves again, later probabilities underflow),
    scores of a magnitude whose exp2 overflows fp32 against a stale reference (inf in the first pass, redone), and
    scores so small that every probability is 1.  Every output row against the fp32 oracle, the usual 2^-7 bar."""
    B, H, N = 8, 16, 2048                         # 1024 workgroups of 256 rows, 32 full key tiles: the ring kernel
    C = H * 64
    g = torch.Generator().manual_seed(hash(pattern) % 1000 + (dtype == torch.float16))
    q = torch.randn(B, N, C, generator=g)
    k = torch.randn(B, N, C, generator=g)
    v = torch.randn(B, N, C, generator=g)
    if pattern == "steady_growth":
        # key norms grow along the sequence: for queries aligned with the growth direction every few tiles hold a new
        # maximum that is 2^8 and more above the last reference
        ramp = torch.linspace(0.2, 6.0, N)[None, :, None]
        direction = torch.randn(1, 1, C, generator=g).sign()
        k = k * 0.3 + direction * ramp
        q = q * 0.3 + direction * (torch.rand(B, N, 1, generator=g) * 2.0 - 0.5)   # some rows aligned, some opposed
    elif pattern == "first_tile_dominates":
        k[:, 5] = q[:, 7] * 6.0                   # key 5 (tile 0) towers over everything for query 7 ...
        k[:, 40] = q[:, 300] * 6.0                # ... key 40 for query 300 (another row block)
    elif pattern == "huge_scores":
        q = q * 6.0
        k = k * 6.0                               # |score * scale| up to ~500: exp2 against a stale reference is inf
    else:
        q = q * 1e-3
        k = k * 1e-3
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), H)
    assert torch.isfinite(got).all()
    assert _attn_err_over_bar(got.cpu().float(), q, k, v, H) < 1.0


@pytest.mark.parametrize("B,H,Nq,Nk,dtype", [(13, 20, 1024, 1024, torch.bfloat16), (8, 16, 2048, 1024, torch.float16),
                                             (32, 16, 512, 1280, torch.bfloat16), (16, 20, 1024, 256, torch.bfloat16),
                                             (32, 16, 512, 320, torch.float16)],
                         ids=["1040wgs", "1024wgs_fp16", "1024wgs_short", "cross_4_tiles", "cross_5_tiles_fp16"])
def test_attention_64_row_kernel_more_workgroups_than_resident_slots(ops, B, H, Nq, Nk, dtype):
    # The ring form of the 64-row kernel (full key tiles; 16, 20, 4 and 5 tiles: every remainder of the three-buffer
    # unrolling).  At least two rounds of the 512 resident (batch, head, 256-row block) workgroups (two per CU): every output
    # row against the fp32 oracle (per batch, to bound host memory), strided q / k / v views of one fused projection,
    # a late score spike in one batch.  (Written for a persistent form of the kernel that was measured and not kept,
    # DESIGN.md section 4; the shapes stay as coverage of multi-round launches.)
    C = H * 64
    g = torch.Generator().manual_seed(B + Nq)
    if Nq == Nk:
        qkv = (torch.randn(B, Nq, 3 * C, generator=g) * 1.3).to(dtype)
        qkv[B - 1, Nk - 40, C:2 * C] = qkv[B - 1, 5, 0:C] * 3.0
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        dq = qkv.to(DEV)
        got = ops.attention(dq[..., :C], dq[..., C:2 * C], dq[..., 2 * C:], H)
    else:
        q = (torch.randn(B, Nq, C, generator=g) * 1.3).to(dtype)
        k = (torch.randn(B, Nk, C, generator=g) * 1.3).to(dtype)
        v = torch.randn(B, Nk, C, generator=g).to(dtype)
        got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), H)
    worst = _attn_err_over_bar(got.cpu().float(), q, k, v, H)
    assert worst < 1.0, worst


def test_attention_full_size_properties(ops):
    # BASELINE config-2 level-1 self-attention (B' = 32, 10 heads, N = 4096): checked through properties.
    B, H, N = 32, 10, 4096
    C = H * 64
    g = torch.Generator(device=DEV).manual_seed(0)
    q = torch.randn(B, N, C, device=DEV, generator=g).to(torch.bfloat16)
    k = torch.randn(B, N, C, device=DEV, generator=g).to(torch.bfloat16)
    ones = torch.ones(B, N, C, device=DEV, dtype=torch.bfloat16)
    # rows of softmax sum to 1: V = 1 -> O = 1 (bf16 rounding of P only)
    o = ops.attention(q, k, ones, H).float()
    assert (o - 1).abs().max().item() < 2.0 ** -7
    # linearity in V
    v1 = torch.randn(B, N, C, device=DEV, generator=g).to(torch.bfloat16)
    v2 = torch.randn(B, N, C, device=DEV, generator=g).to(torch.bfloat16)
    o12 = ops.attention(q, k, (v1.float() + v2.float()).to(torch.bfloat16), H).float()
    o1, o2 = ops.attention(q, k, v1, H).float(), ops.attention(q, k, v2, H).float()
    assert (o12 - (o1 + o2)).abs().max().item() < 0.02
    # one sampled (batch, head) against the oracle
    b, h = 17, 6
    sl = slice(h * 64, (h + 1) * 64)
    want = ops_ref.attention_ref(q[b:b + 1, :, sl].cpu(), k[b:b + 1, :, sl].cpu(), v1[b:b + 1, :, sl].cpu(), 1).float()
    assert (o1[b:b + 1, :, sl].cpu() - want).abs().max().item() < 2.0 ** -7


# ---- transformer-block row kernels -------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 1024, 2 * 5120), (3, 17, 2 * 2560), (1, 1, 16), (4096 + 3, 2 * 40)])
def test_geglu_bit_exact(ops, dtype, shape):
    g = torch.Generator().manual_seed(shape[-1])
    h = (torch.randn(shape, generator=g) * 2.5).to(dtype)
    want = ops_ref.geglu_ref(h)                      # torch CPU: the same two roundings (gelu, product)
    got = ops.geglu(h.to(DEV))
    assert got.shape == want.shape and got.dtype == dtype
    # the device erf (ocml) and the host libm erf may differ by an fp32 ulp before the rounding to 16 bits:
    # bit-exact against torch's own two-op form ON THE DEVICE, <= 1 ulp of the io type against the CPU oracle
    hidden, gate = h.to(DEV).chunk(2, dim=-1)
    dev_ref = hidden * torch.nn.functional.gelu(gate)
    ulp = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    if dtype == torch.bfloat16:      # the reference's dtype (lavit_for_pigeon.py:706 autocast bf16)
        assert torch.equal(got, dev_ref)
    else:                            # f16 keeps 3 more bits: an fp32-ulp difference between erf builds shows up, rarely
        assert (got != dev_ref).float().mean().item() < 1e-3
        assert bool(((got.float() - dev_ref.float()).abs()
                     <= 2 * ulp * dev_ref.float().abs() + 2e-7 * (hidden.float() * gate.float()).abs() + 1e-6).all())
    err = (got.cpu().float() - want.float()).abs()
    # 1 + erf(x) cancels for x << 0, so two fp32 erf implementations (host libm, device ocml) differ there by up to
    # ~1e-7 ABSOLUTE in (1 + erf): |hidden * gate| * 2e-7 covers it (visible in f16 only)
    hid, gat = h.float().chunk(2, dim=-1)
    assert bool((err <= 2 * ulp * want.float().abs() + 2e-7 * (hid * gat).abs() + 1e-6).all())
    assert (got.cpu() != want).float().mean().item() < (0.01 if dtype == torch.bfloat16 else 0.05)


def test_geglu_strided_rows(ops):
    # rows of a wider buffer (leading dimension > 2 D): no copy, same result
    big = torch.randn(64, 3 * 256, device=DEV).to(torch.bfloat16)
    h = big[:, :512]
    hidden, gate = h.chunk(2, dim=-1)
    assert torch.equal(ops.geglu(h), hidden * torch.nn.functional.gelu(gate))


LN_CASES = [(2, 1024, 1280), (2, 4096, 640), (5, 7, 320), (3, 1, 8), (2, 33, 2560), (1, 9, 4096), (2, 5, 1408)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", LN_CASES, ids=[f"{s[0]}x{s[1]}x{s[2]}" for s in LN_CASES])
@pytest.mark.parametrize("with_res", [False, True])
def test_add_layernorm_matches_oracle(ops, dtype, shape, with_res):
    g = torch.Generator().manual_seed(shape[-1] + shape[1])
    C = shape[-1]
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3 * torch.randn(1, 1, C, generator=g)).to(dtype)
    r = (torch.randn(shape, generator=g) * 0.7).to(dtype) if with_res else None
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    b = (0.2 * torch.randn(C, generator=g)).to(dtype)
    want_s, want_y, want32 = ops_ref.add_layer_norm_ref(x, r, w, b, 1e-5)
    got_s, got_y = ops.add_layer_norm(x.to(DEV), None if r is None else r.to(DEV), w.to(DEV), b.to(DEV), 1e-5)
    assert torch.equal(got_s.cpu(), want_s)          # the residual sum is one exact fp32 add + one rounding
    ulp = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    err = (got_y.cpu().float() - want32).abs()
    # fp32 statistics both sides, one rounding to the io type: within 1 ulp of the io type of the fp32 oracle
    assert bool((err <= ulp * want32.abs() + 4 * ulp * 2.0 ** -7).all()), err.max().item()
    assert (got_y.cpu() != want_y).float().mean().item() < 0.02


def test_add_layernorm_no_affine_and_fp32_params(ops):
    x = torch.randn(4, 10, 640).to(torch.bfloat16)
    _, y = ops.add_layer_norm(x.to(DEV), None, None, None, 1e-6)
    want = torch.nn.functional.layer_norm(x.float(), (640,), None, None, 1e-6)
    assert (y.cpu().float() - want).abs().max().item() < 2.0 ** -6
    w, b = torch.rand(640) + 0.5, torch.randn(640)
    _, y = ops.add_layer_norm(x.to(DEV), None, w.to(DEV), b.to(DEV), 1e-5)
    want = torch.nn.functional.layer_norm(x.float(), (640,), w, b, 1e-5)
    assert (y.cpu().float() - want).abs().max().item() < 2.0 ** -5


def test_row_kernels_full_size_properties(ops):
    # BASELINE config-2 sizes (B' = 32, level-2 tokens: M = 32768 rows): size-independent properties.
    M, C = 32 * 1024, 1280
    x = (torch.randn(M, C, device=DEV) * 2 + 0.5).to(torch.bfloat16)
    r = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    s, y = ops.add_layer_norm(x, r, None, None, 1e-5)
    assert torch.equal(s, x + r)
    yf = y.float()
    assert yf.mean(-1).abs().max().item() < 4e-3 and (yf.var(-1, unbiased=False) - 1).abs().max().item() < 1e-2
    h = torch.randn(M, 2 * 4 * C, device=DEV).to(torch.bfloat16)
    out = ops.geglu(h)
    # gate -> +large: gelu(g) = g, so geglu(h) = hidden * gate there; gate = 0 -> 0
    h2 = h.clone()
    h2[:, 4 * C:] = 0
    assert ops.geglu(h2).abs().max().item() == 0
    hidden, gate = h.chunk(2, dim=-1)
    assert torch.equal(out, hidden * torch.nn.functional.gelu(gate))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 3, 64, 48), (1, 3, 7, 5), (1, 1, 16, 16), (3, 4, 9, 4)])
def test_image_to_uint8_bit_exact(ops, dtype, shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 0.8).to(dtype)
    # exact ties of the final rounding and the clamp edges
    x.view(-1)[:6] = torch.tensor([-1.0, 1.0, -3.0, 3.0, 1.0 / 255 - 1, 0.0]).to(dtype)
    # lavit_for_pigeon.py:780-782 + numpy_to_pil :929
    ref = (x.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    want = (ref * 255).round().astype("uint8")
    got = ops.image_to_uint8(x.to(DEV)).cpu().numpy()
    assert got.shape == want.shape and got.dtype == want.dtype
    assert (got == want).all()


# (B, heads, Nq, Nk, head_dim, bias): head dims beyond the UNet's 64
ATTN_D_CASES = [
    (2, 16, 256, 256, 88, False),    # token decoder self-attention (modeling_decoder.py:44-71): 16 heads x 88
    (2, 16, 256, 90, 88, True),      # token decoder cross-attention with the -10000 pad mask (:74-109)
    (2, 4, 300, 300, 128, False),
    (1, 2, 4096, 4096, 128, False),
    (2, 8, 1024, 77, 40, False),     # SD-1.5 level 1 (lowres variant, lavit_for_pigeon.py:822-895)
    (2, 8, 256, 77, 80, True),       # SD-1.5 level 2 with encoder_attention_mask
    (1, 3, 65, 130, 8, False),
    (1, 2, 130, 65, 120, True),
    (2, 8, 64, 77, 160, True),       # SD-1.5 levels 3 / mid: 8 heads x 160 (one wavefront per SIMD instantiation)
    (2, 8, 256, 256, 160, False),
    (1, 2, 300, 300, 256, False),
]


@pytest.mark.parametrize("case", ATTN_D_CASES, ids=[f"B{c[0]}H{c[1]}q{c[2]}k{c[3]}d{c[4]}{'b' if c[5] else ''}" for c in ATTN_D_CASES])
def test_attention_other_head_dims(ops, case):
    B, H, Nq, Nk, D, has_bias = case
    g = torch.Generator().manual_seed(Nq + Nk + D)
    C = H * D
    q = (torch.randn(B, Nq, C, generator=g) * 1.2).to(torch.bfloat16)
    k = (torch.randn(B, Nk, C, generator=g) * 1.2).to(torch.bfloat16)
    v = torch.randn(B, Nk, C, generator=g).to(torch.bfloat16)
    bias = None
    if has_bias:
        keep = torch.randint(1, Nk + 1, (B,), generator=g)
        bias = (1.0 - (torch.arange(Nk)[None, :] < keep[:, None]).float()) * -10000.0
    want = ops_ref.attention_ref(q, k, v, H, None, bias).float()
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), H, None, None if bias is None else bias.to(DEV)).cpu().float()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < 2.0 ** -7 * max(1.0, want.abs().max().item())
    with pytest.raises(NotImplementedError):
        ops.attention(torch.zeros(1, 8, 512, device=DEV).bfloat16(), torch.zeros(1, 8, 512, device=DEV).bfloat16(),
                      torch.zeros(1, 8, 512, device=DEV).bfloat16(), 1)   # the VAE's 512-wide head is not covered


NHWC_CASES = [
    (2, 320, 128, 128, True, True), (2, 640, 64, 64, True, False), (2, 1280, 32, 32, True, True),
    (2, 2560, 32, 32, True, False), (1, 960, 128, 128, True, False), (2, 640, 64, 64, False, False),
    (2, 640, 76, 52, True, True), (2, 1920, 38, 26, True, False), (1, 128, 256, 256, True, False),
    (3, 64, 5, 3, True, True), (2, 512, 16, 16, False, False), (1, 256, 1, 1, True, False),
]


@pytest.mark.parametrize("case", NHWC_CASES, ids=[f"{c[1]}x{c[2]}x{c[3]}{'s' if c[4] else ''}{'a' if c[5] else ''}" for c in NHWC_CASES])
def test_groupnorm_channels_last_matches_oracle(ops, case):
    N, C, H, W, silu, with_add = case
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(N, C, H, W, generator=g) * (0.5 + torch.rand(1, C, 1, 1, generator=g) * 2)
         + torch.randn(1, C, 1, 1, generator=g)).to(torch.bfloat16)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(torch.bfloat16)
    b = (0.2 * torch.randn(C, generator=g)).to(torch.bfloat16)
    t = (torch.randn(N, C, generator=g) * 0.5).to(torch.bfloat16) if with_add else None
    eps = 1e-5 if silu else 1e-6
    xin = x if t is None else x + t[:, :, None, None]
    want = ops_ref.group_norm_silu_ref(xin, 32, w, b, eps, silu)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    got = ops.group_norm_silu(xd, 32, w.to(DEV), b.to(DEV), eps, silu, add_nc=None if t is None else t.to(DEV))
    assert got.shape == x.shape and got.dtype == x.dtype
    if H * W > 1:
        assert got.is_contiguous(memory_format=torch.channels_last)
    err = (got.cpu().float() - want.float()).abs()
    ulp = 2.0 ** -7
    assert bool((err <= ulp * want.float().abs() + 4 * ulp * 2.0 ** -7 + 1e-6).all()), err.max().item()
    assert (got.cpu() != want).float().mean().item() < 0.02
    again = ops.group_norm_silu(xd, 32, w.to(DEV), b.to(DEV), eps, silu, add_nc=None if t is None else t.to(DEV))
    assert torch.equal(got, again)     # no atomics: run-to-run bit-identical


@pytest.mark.parametrize("shape", [(2, 1280, 32, 32), (1, 640, 64, 64), (3, 40, 5, 3), (1, 2560, 3, 7), (2, 8, 1, 1),
                                   (1, 512, 128, 128), (1, 24, 1, 300)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_upsample_nearest2x_channels_last_is_bit_exact(ops, shape, dtype):
    # diffusers Upsample2D.forward: F.interpolate(x, scale_factor=2.0, mode="nearest"); pure data movement
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(dtype)
    want = torch.nn.functional.interpolate(x.float(), scale_factor=2.0, mode="nearest").to(dtype)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    got = ops.upsample_nearest2x(xd)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got.cpu(), want)
    if shape[1] > 1 and shape[2] * shape[3] > 1:
        with pytest.raises(ValueError):
            ops.upsample_nearest2x(x.to(DEV))          # NCHW memory
    with pytest.raises(TypeError):
        ops.upsample_nearest2x(xd.float())


CAT_CASES = [(2, 1280, 1280, 32, 32), (2, 1280, 640, 32, 32), (1, 640, 320, 64, 64), (2, 320, 320, 38, 26),
             (3, 40, 24, 5, 3), (1, 8, 248, 1, 1)]


@pytest.mark.parametrize("case", CAT_CASES, ids=[f"{c[1]}+{c[2]}x{c[3]}x{c[4]}" for c in CAT_CASES])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_groupnorm_over_an_unwritten_concatenation(ops, case, dtype):
    # the up blocks' torch.cat([hidden, skip], dim=1) + norm1: both halves read in place; groups straddle the boundary
    # in the 1280+640 / 640+320 / 40+24 cases (60, 30, 2 channels per group).  Same partial sums in the same order as
    # the one-source kernel on the materialised concatenation -> bit-identical to it, and within the oracle's bar
    N, C1, C2, H, W = case
    g = torch.Generator().manual_seed(C1 + C2 + H)
    x1 = (torch.randn(N, C1, H, W, generator=g) * 1.5 + torch.randn(1, C1, 1, 1, generator=g)).to(dtype)
    x2 = (torch.randn(N, C2, H, W, generator=g) * 0.7 - 0.5).to(dtype)
    C = C1 + C2
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    b = (0.2 * torch.randn(C, generator=g)).to(dtype)
    d1 = x1.to(DEV).contiguous(memory_format=torch.channels_last)
    d2 = x2.to(DEV).contiguous(memory_format=torch.channels_last)
    for silu in (True, False):
        got = ops.group_norm_silu_cat(d1, d2, 32 if C % 32 == 0 else 8, w.to(DEV), b.to(DEV), 1e-5, silu)
        cat = torch.cat([d1, d2], dim=1).contiguous(memory_format=torch.channels_last)
        two_step = ops.group_norm_silu(cat, 32 if C % 32 == 0 else 8, w.to(DEV), b.to(DEV), 1e-5, silu)
        assert got.shape == (N, C, H, W) and torch.equal(got, two_step)
        if H * W > 1:
            assert got.is_contiguous(memory_format=torch.channels_last)
        want = ops_ref.group_norm_silu_ref(torch.cat([x1, x2], dim=1), 32 if C % 32 == 0 else 8, w, b, 1e-5, silu)
        err = (got.cpu().float() - want.float()).abs()
        ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        assert bool((err <= ulp * want.float().abs() + 4 * ulp * 2.0 ** -7 + 1e-6).all()), err.max().item()
    with pytest.raises(ValueError):
        ops.group_norm_silu_cat(d1, d2.contiguous() if H * W > 1 and C2 > 1 else d2[:, :, :0], 32, None, None)
    with pytest.raises(ValueError):
        ops.group_norm_silu_cat(d1, d2[:1].expand(N + 1, -1, -1, -1), 32, None, None)


@pytest.mark.parametrize("shape", [(2, 320, 16, 24), (3, 64, 5, 7), (1, 1280, 8, 8), (2, 8, 1, 3)])
@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_bias_residual_add_is_bit_exact_in_both_layouts(ops, shape, with_res, dtype):
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 3).to(dtype)
    res = (torch.randn(*shape, generator=g) * 3).to(dtype) if with_res else None
    for bdt in (dtype, torch.float32):
        bias = torch.randn(shape[1], generator=g).to(bdt)
        want = ops_ref.bias_residual_add_ref(x, bias, res)
        # channels_last memory (inner = 1)
        xc = x.to(DEV).contiguous(memory_format=torch.channels_last)
        rc = None if res is None else res.to(DEV).contiguous(memory_format=torch.channels_last)
        got = ops.bias_residual_add(xc, bias.to(DEV), rc, inplace=False)
        assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got.cpu(), want)
        if (shape[2] * shape[3]) % 8 == 0:     # NCHW memory (inner = H*W)
            got = ops.bias_residual_add(x.to(DEV), bias.to(DEV), None if res is None else res.to(DEV), inplace=False)
            assert torch.equal(got.cpu(), want)
    # in place by default: the convolution output is overwritten
    xi = x.to(DEV).contiguous(memory_format=torch.channels_last)
    out = ops.bias_residual_add(xi, bias.to(DEV), None)
    assert out.data_ptr() == xi.data_ptr()


def test_bias_residual_add_rejects_unvectorisable_shapes(ops):
    x = torch.zeros(1, 4, 3, 3, device=DEV, dtype=torch.bfloat16)     # NCHW with H*W % 8 != 0
    with pytest.raises(NotImplementedError):
        ops.bias_residual_add(x, torch.zeros(4, device=DEV, dtype=torch.bfloat16))
    with pytest.raises(ValueError):
        ops.bias_residual_add(x[:, :, :, :2], torch.zeros(4, device=DEV, dtype=torch.bfloat16))


@pytest.mark.parametrize("shape", [(5, 16384), (3, 1000), (2, 7), (4, 4100), (1, 1)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_softmax_rows_matches_fp32_softmax(ops, shape, dtype):
    g = torch.Generator().manual_seed(sum(shape))
    s = torch.randn(*shape, generator=g) * 20.0
    if shape[1] > 100:
        s[0, 17] = 300.0                     # a spike: everything else underflows, no NaN
    got = ops.softmax_rows(s.to(DEV), 0.25, dtype)
    want = torch.softmax(s * 0.25, dim=-1)
    assert got.dtype == dtype and not torch.isnan(got.float()).any()
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    assert (got.float().cpu() - want).abs().max().item() <= ulp * want.max().item() + 1e-7   # one rounding of P
    assert torch.allclose(got.float().sum(-1).cpu(), torch.ones(shape[0]), atol=4 * ulp)


def test_wide_head_attention_matches_oracle(ops):
    # the VAE's head shape (one 512-wide head), ragged lengths, two query chunks
    for Nq, Nk, D in ((300, 260, 512), (1024, 1024, 512), (64, 128, 320)):
        g = torch.Generator().manual_seed(Nq + D)
        q = torch.randn(1, Nq, D, generator=g).bfloat16()
        k = torch.randn(1, Nk, D, generator=g).bfloat16()
        v = torch.randn(1, Nk, D, generator=g).bfloat16()
        want = ops_ref.attention_ref(q, k, v, 1, D ** -0.5, None).float()[0]
        got = ops.wide_head_attention(q[0].to(DEV), k[0].to(DEV), v[0].to(DEV), D ** -0.5, rows_per_chunk=256).float().cpu()
        err = (got - want).abs().max().item()
        assert err < 2.0 ** -7 * max(1.0, want.abs().max().item()), (Nq, Nk, D, err)


@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("shape", [(2, 100, 1280), (3, 7, 640), (1, 5, 8)])
def test_add_layernorm_with_pending_column_bias(ops, shape, with_res):
    # pgn_add_layernorm_cb: the bias owed by a bias-free accumulating projection enters as an fp32 [C] vector
    g = torch.Generator().manual_seed(shape[1])
    C = shape[-1]
    x = (torch.randn(shape, generator=g) * 1.5).bfloat16()
    r = (torch.randn(shape, generator=g) * 0.7).bfloat16() if with_res else None
    cb = torch.randn(C, generator=g) * 0.5
    w = (1 + 0.2 * torch.randn(C, generator=g)).bfloat16()
    b = (0.2 * torch.randn(C, generator=g)).bfloat16()
    want_s, _, want32 = ops_ref.add_layer_norm_ref(x, r, w, b, 1e-5, col_bias=cb)
    got_s, got_y = ops.add_layer_norm(x.to(DEV), None if r is None else r.to(DEV), w.to(DEV), b.to(DEV), 1e-5,
                                      col_bias=cb.to(DEV))
    if with_res:
        assert (got_s.cpu() != want_s).float().mean().item() < 0.01      # x + cb + r: fp32 add order may differ
        assert (got_s.cpu().float() - want_s.float()).abs().max() <= 2.0 ** -7 * want_s.float().abs().max()
    else:
        assert got_s.data_ptr() == got_s.data_ptr() and torch.equal(got_s.cpu(), x)   # the bias stays pending
    ulp = 2.0 ** -7
    err = (got_y.cpu().float() - want32).abs()
    assert bool((err <= 2 * ulp * want32.abs() + 8 * ulp * 2.0 ** -7).all()), err.max().item()
    # and it matters
    _, plain = ops.add_layer_norm(x.to(DEV), None if r is None else r.to(DEV), w.to(DEV), b.to(DEV), 1e-5)
    assert (plain.cpu().float() - want32).abs().max() > 0.05
