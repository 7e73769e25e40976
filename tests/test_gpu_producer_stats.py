"""GroupNorm statistics emitted by the launch that PRODUCES the tensor (first alternative):
`pgn_bias_residual_add_stats` + `pgn_groupnorm_silu_nhwc_pre` (pigeon_amd/csrc/groupnorm_nhwc.hip:
pgn_bias_res_stats_kernel; call sites: ResnetBlock2D's `input_tensor + hidden_states`, the convolutions' bias adds and
Transformer2DModel's `hidden_states + residual` of diffusers 0.27.2, reached from lavit_for_pigeon.py:752-757).

What has to hold, all through the C ABI:
  * y is bit-identical to pgn_bias_residual_add's;
  * the partial words are bit-identical to the ones the statistics launch of pgn_groupnorm_silu_nhwc writes after
    reading y back (same chunking, same pivots, same order), so
  * pgn_groupnorm_silu_nhwc_pre on a single input is bit-identical to the three-launch form, and on a concatenation
    (two producers, two chunkings) within one unit in the last place of it and inside the fp64 oracle's bar;
  * y aliasing an input is refused (every workgroup of a sample reads the pivot rows of the inputs);
and through the modules: a ResnetBlock2D chain / an up-block concatenation / a Transformer2DModel give the same
tensors with layers.PRODUCER_STATS on and off, and the partials do not survive a write to the tensor."""
import pytest
import torch

from oracle import ops_ref
from pigeon_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dt(t):
    return _lib.PGN_BF16 if t.dtype == torch.bfloat16 else _lib.PGN_F16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rows(N, HW, C, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(N, HW, C, generator=g) * (0.5 + 2 * torch.rand(1, 1, C, generator=g))
         + 2 * torch.randn(1, 1, C, generator=g)) * scale
    return x.to(dtype).to(DEV)


def _produce(lib, x, res, bias):
    N, HW, C = x.shape
    nb = int(lib.pgn_bias_residual_add_stats_bytes(N, C, HW))
    assert nb > 0 and nb % (N * C * 8) == 0
    part = torch.full((nb,), 0xA5, dtype=torch.uint8, device=DEV)
    y = torch.full_like(x, float("nan"))
    rc = lib.pgn_bias_residual_add_stats(x.data_ptr(), None if res is None else res.data_ptr(), bias.data_ptr(), _dt(bias),
                                         y.data_ptr(), _dt(x), N, C, HW, part.data_ptr(), nb, _stream())
    assert rc == 0, lib.pgn_error_string(rc)
    return y, part


def _plain(lib, x, res, bias):
    N, HW, C = x.shape
    y = torch.empty_like(x)
    rc = lib.pgn_bias_residual_add(x.data_ptr(), None if res is None else res.data_ptr(), bias.data_ptr(), _dt(bias),
                                   y.data_ptr(), _dt(x), N * HW, C, 1, _stream())
    assert rc == 0
    return y


def _three_pass(lib, x, gamma, beta, G, eps, silu, x2=None):
    N, HW, C1 = x.shape
    C = C1 + (0 if x2 is None else x2.shape[2])
    nb = int(lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, G))
    ws = torch.full((nb,), 0xA5, dtype=torch.uint8, device=DEV)
    y = torch.empty((N, HW, C), dtype=x.dtype, device=DEV)
    act = (1 if silu else 0) | 2          # PGN_GN_ONEPASS_NEVER: the three launches whatever the size
    if x2 is None:
        rc = lib.pgn_groupnorm_silu_nhwc(x.data_ptr(), y.data_ptr(), _dt(x), gamma.data_ptr(), beta.data_ptr(), _dt(gamma),
                                         None, _lib.PGN_F32, N, C, HW, G, eps, act, ws.data_ptr(), nb, _stream())
    else:
        rc = lib.pgn_groupnorm_silu_nhwc_cat(x.data_ptr(), C1, x2.data_ptr(), C - C1, y.data_ptr(), _dt(x),
                                             gamma.data_ptr(), beta.data_ptr(), _dt(gamma), None, _lib.PGN_F32, N, HW, G,
                                             eps, act, ws.data_ptr(), nb, _stream())
    assert rc == 0, lib.pgn_error_string(rc)
    return y, ws


def _pre(lib, x, part, gamma, beta, G, eps, silu, x2=None, part2=None):
    N, HW, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[2]
    ws = torch.full((N * G * 8,), 0xA5, dtype=torch.uint8, device=DEV)
    y = torch.full((N, HW, C1 + C2), float("nan"), dtype=x.dtype, device=DEV)
    rc = lib.pgn_groupnorm_silu_nhwc_pre(x.data_ptr(), C1, part.data_ptr(), None if x2 is None else x2.data_ptr(), C2,
                                         None if part2 is None else part2.data_ptr(), y.data_ptr(), _dt(x),
                                         gamma.data_ptr(), beta.data_ptr(), _dt(gamma), N, HW, G, eps, 1 if silu else 0,
                                         ws.data_ptr(), ws.numel(), _stream())
    assert rc == 0, lib.pgn_error_string(rc)
    return y


def _affine(C, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return ((1 + 0.2 * torch.randn(C, generator=g)).to(dtype).to(DEV), (0.2 * torch.randn(C, generator=g)).to(dtype).to(DEV))


def _oracle(x, gamma, beta, G, eps, silu):
    N, HW, C = x.shape
    nchw = x.cpu().permute(0, 2, 1).reshape(N, C, HW, 1)
    return ops_ref.group_norm_silu_ref(nchw, G, gamma.cpu(), beta.cpu(), eps, silu).reshape(N, C, HW).permute(0, 2, 1)


def _ulp_apart(a, b, floor=2e-6):
    """largest |a - b| in units of the last place of the larger of the two (16-bit type); differences below `floor`
    count as zero: where (x - mean) * rstd * gamma + beta cancels to ~0 an fp32-rounding change of the mean moves the
    result by many of ITS last places while staying ~1e-7 in absolute terms"""
    fa, fb = a.float(), b.float()
    d = (fa - fb).abs()
    big = torch.maximum(fa.abs(), fb.abs())
    mant = 8 if a.dtype == torch.bfloat16 else 11
    ulp = torch.exp2(torch.floor(torch.log2(big.clamp_min(1e-30))) - (mant - 1))
    d = torch.where(d <= floor, torch.zeros_like(d), d)
    return float((d / ulp).max())


CASES = [
    # N, HW, C, G, residual, silu, dtype
    (4, 128 * 128, 320, 32, True, True, torch.bfloat16),     # the bench's largest level (42 MB here, 336 MB there)
    (4, 64 * 64, 640, 32, False, False, torch.bfloat16),     # Transformer2DModel.norm: no activation
    (6, 38 * 26, 1280, 32, True, True, torch.float16),       # ragged rows (1216x832 / 304x208)
    (2, 32 * 32, 2560, 32, False, True, torch.bfloat16),     # one row lane per workgroup
    (5, 77, 64, 32, True, True, torch.float16),              # a lane's 8 channels span 4 groups
    (2, 1, 256, 32, False, False, torch.bfloat16),           # one row: the three pivot rows coincide
    (3, 300, 32, 32, True, True, torch.bfloat16),            # one channel per group
]


@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}hw{c[1]}c{c[2]}{'r' if c[4] else ''}{'s' if c[5] else ''}" for c in CASES])
def test_producer_words_are_the_statistics_launch_words(case):
    N, HW, C, G, with_res, silu, dtype = case
    lib = _lib.load()
    x = _rows(N, HW, C, dtype, C + HW)
    res = _rows(N, HW, C, dtype, C + HW + 1) if with_res else None
    bias = (0.3 * torch.randn(C, generator=torch.Generator().manual_seed(C))).to(dtype).to(DEV)
    gamma, beta = _affine(C, dtype, 5)
    y, part = _produce(lib, x, res, bias)
    assert torch.equal(y, _plain(lib, x, res, bias))
    want, ws = _three_pass(lib, y, gamma, beta, G, 1e-5, silu)
    # the workspace of the three-launch form starts with its [N][S][C] partial words
    assert torch.equal(ws[:part.numel()].view(torch.int64), part.view(torch.int64))
    got = _pre(lib, y, part, gamma, beta, G, 1e-5, silu)
    assert torch.equal(got, want)
    ref = _oracle(y, gamma, beta, G, 1e-5, silu)
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert ((got.float().cpu() - ref).abs() <= tol * (1 + ref.abs())).all()


@pytest.mark.parametrize("shape", [(4, 64 * 64, 640, 320, torch.bfloat16), (3, 38 * 26, 1280, 640, torch.float16),
                                   (2, 32 * 32, 1280, 1280, torch.bfloat16), (4, 128 * 128, 320, 320, torch.bfloat16)],
                         ids=["640+320", "1280+640r", "1280+1280", "320+320"])
def test_concatenation_of_two_producers(shape):
    N, HW, C1, C2, dtype = shape
    lib = _lib.load()
    G = 32
    bias1 = torch.zeros(C1, dtype=dtype, device=DEV) + 0.25
    bias2 = torch.zeros(C2, dtype=dtype, device=DEV) - 0.5
    y1, p1 = _produce(lib, _rows(N, HW, C1, dtype, 1), _rows(N, HW, C1, dtype, 2), bias1)
    y2, p2 = _produce(lib, _rows(N, HW, C2, dtype, 3, scale=3.0), None, bias2)
    gamma, beta = _affine(C1 + C2, dtype, 9)
    want, _ = _three_pass(lib, y1, gamma, beta, G, 1e-5, True, x2=y2)
    got = _pre(lib, y1, p1, gamma, beta, G, 1e-5, True, x2=y2, part2=p2)
    apart = _ulp_apart(got, want)
    frac = (got != want).float().mean().item()
    print(f"cat {C1}+{C2} {dtype}: {apart:.2f} ulp apart at most, {100 * frac:.3f} % of the elements differ")
    assert apart <= 1 and frac < 0.02
    ref = _oracle(torch.cat([y1, y2], dim=2), gamma, beta, G, 1e-5, True)
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert ((got.float().cpu() - ref).abs() <= tol * (1 + ref.abs())).all()


def test_argument_checks():
    lib = _lib.load()
    N, HW, C = 2, 256, 64
    x, res = _rows(N, HW, C, torch.bfloat16, 1), _rows(N, HW, C, torch.bfloat16, 2)
    bias = torch.zeros(C, dtype=torch.bfloat16, device=DEV)
    nb = int(lib.pgn_bias_residual_add_stats_bytes(N, C, HW))
    part = torch.empty(nb, dtype=torch.uint8, device=DEV)
    args = lambda xx, rr, yy, pb=nb: lib.pgn_bias_residual_add_stats(
        xx.data_ptr(), None if rr is None else rr.data_ptr(), bias.data_ptr(), _lib.PGN_BF16, yy.data_ptr(), _lib.PGN_BF16,
        N, C, HW, part.data_ptr(), pb, _stream())
    y = torch.empty_like(x)
    assert args(x, res, x) == _lib.PGN_ERR_INVALID          # in place over x
    assert args(x, res, res) == _lib.PGN_ERR_INVALID        # in place over the residual
    assert args(x, res, y, nb - 8) == _lib.PGN_ERR_INVALID  # partials buffer too small
    assert args(x, res, y) == 0
    assert int(lib.pgn_bias_residual_add_stats_bytes(N, 60, HW)) == 0       # C % 8
    gamma, beta = _affine(C, torch.bfloat16, 1)
    ws = torch.empty(N * 32 * 8, dtype=torch.uint8, device=DEV)
    pre = lambda p1, x2, c2, p2, act=1, wsb=ws.numel(): lib.pgn_groupnorm_silu_nhwc_pre(
        y.data_ptr(), C, p1, x2, c2, p2, torch.empty_like(y).data_ptr(), _lib.PGN_BF16, gamma.data_ptr(), beta.data_ptr(),
        _lib.PGN_BF16, N, HW, 32, 1e-5, act, ws.data_ptr(), wsb, _stream())
    assert pre(None, None, 0, None) == _lib.PGN_ERR_INVALID                 # no partials
    assert pre(part.data_ptr(), y.data_ptr(), C, None) == _lib.PGN_ERR_INVALID   # second half without partials
    assert pre(part.data_ptr(), None, 0, None, act=3) == _lib.PGN_ERR_INVALID    # form bits have no meaning here
    assert pre(part.data_ptr(), None, 0, None, wsb=N * 32 * 8 - 8) == _lib.PGN_ERR_INVALID
    assert pre(part.data_ptr(), None, 0, None) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_modules_agree_with_and_without_producer_statistics(monkeypatch, dtype):
    """Two ResnetBlock2Ds (identity shortcut, then channel change over a concatenation) and a Transformer2DModel at the
    UNet's level-1 width: PRODUCER_STATS on against off.  GN_STATS_MIN_BYTES = 0 so that the small tensors of a test
    take the path the bench's > 32 MiB tensors take."""
    from pigeon_amd import layers, ops
    from pigeon_amd.unet import Transformer2DModel
    torch.manual_seed(3)
    N, C, H, W = 2, 640, 24, 20
    r1 = layers.ResnetBlock2D(C, C, 1280).to(DEV, dtype).eval()
    r2 = layers.ResnetBlock2D(C + 320, C, 1280).to(DEV, dtype).eval()
    tr = Transformer2DModel(10, 64, C, 1, 2048, 32, True).to(DEV, dtype).eval()
    down = layers.Downsample2D(320).to(DEV, dtype).eval()
    x = torch.randn(N, C, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    big = torch.randn(N, 320, 2 * H, 2 * W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    temb = torch.randn(N, 1280, device=DEV).to(dtype)
    ctx = torch.randn(N, 77, 2048, device=DEV).to(dtype)

    def run():
        with torch.no_grad():
            skip = down(big)                                   # a convolution's bias add as producer
            h = r1(x, temb)                                    # identity shortcut: bias + residual as producer
            h = tr(h, encoder_hidden_states=ctx)               # norm consumes r1's partials; output add as producer
            return r2(h, temb, skip=skip), h, skip             # concatenation of two producers

    monkeypatch.setattr(ops, "GN_STATS_MIN_BYTES", 0)
    calls = {"pre": 0}
    lib = _lib.load()
    real = lib.pgn_groupnorm_silu_nhwc_pre

    class _Count:
        def __call__(self, *a):
            calls["pre"] += 1
            return real(*a)
    monkeypatch.setattr(lib, "pgn_groupnorm_silu_nhwc_pre", _Count())
    on = run()
    assert calls["pre"] == 2, calls          # tr.norm and r2.norm1 (the concatenation); r1.norm1's input has no producer here
    assert ops.gn_partials(on[1]) is not None and ops.gn_partials(on[2]) is not None
    monkeypatch.setattr(layers, "PRODUCER_STATS", False)
    calls["pre"] = 0
    off = run()
    assert calls["pre"] == 0 and ops.gn_partials(off[1]) is None
    for a, b, name in zip(on, off, ("resnet over the concatenation", "transformer", "downsample")):
        rel = ((a.float() - b.float()).norm() / b.float().norm()).item()
        print(f"{dtype} {name}: PRODUCER_STATS on vs off relL2 {rel:.2e}")
        assert rel < (0.01 if dtype == torch.bfloat16 else 0.0015), (name, rel)
    assert torch.equal(on[2], off[2])                          # same bias add, bit for bit
    # partials do not survive a write to the tensor
    t = on[1]
    t.add_(1)
    assert ops.gn_partials(t) is None
