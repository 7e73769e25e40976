"""GPU tests of the one-pass channels-last GroupNorm (pigeon_amd/csrc/groupnorm_nhwc.hip: pgn_gn_nhwc_onepass_kernel).

The form keeps a workgroup's rows in registers and exchanges per-group partial sums between the workgroups of a sample
inside one launch, so what has to be shown beyond the oracle bar of test_gpu_kernels.py is the exchange itself:
  * it agrees with the three-pass form (PGN_GN_NHWC_ONEPASS=0) to the last-bit noise of two summation orders;
  * the bounded-wait safety net (every workgroup recomputes its peers' words: PGN_GN_NHWC_ONEPASS_TIMEOUT=0) produces
    the same bits as the exchange;
  * a workspace that still holds another call's words and counters (and whose lines the CUs have just read) does not
    leak into the next call: back-to-back calls on ONE workspace, every element checked;
  * the full bench shape (every CU streaming, several rounds of workgroups) gives the same bits on every run.
By default the library takes the form only for tensors of up to 32 MiB (it is slower than three passes beyond, see
groupnorm_nhwc.hip: nh_run); PGN_GN_NHWC_ONEPASS=1 (mapped onto the `act` bit PGN_GN_ONEPASS_ALWAYS by
pigeon_amd.ops) takes it wherever the shape allows, which is what these tests set.
"""
import contextlib
import os

import pytest
import torch

from oracle import ops_ref
from pigeon_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@contextlib.contextmanager
def _env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    try:
        for k, v in kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _call(lib, x, y, gamma, beta, add, G, eps, act, ws, dt=None, x2=None):
    """x: [N, HW, C1] device rows (x2: [N, HW, C2] or None); launches on the current stream, no synchronisation."""
    N, HW, C1 = x.shape
    dt = dt if dt is not None else (_lib.PGN_BF16 if x.dtype == torch.bfloat16 else _lib.PGN_F16)
    # the library reads no environment: the form travels in `act` (PGN_GN_ONEPASS_* bits); the switches this file
    # flips with _env are mapped onto them where the Python binding does it
    from pigeon_amd import ops
    act = ops._gn_nhwc_act(bool(act))
    s = torch.cuda.current_stream().cuda_stream
    ap = None if add is None else add.data_ptr()
    if x2 is None:
        rc = lib.pgn_groupnorm_silu_nhwc(x.data_ptr(), y.data_ptr(), dt, gamma.data_ptr(), beta.data_ptr(), dt, ap, dt,
                                         N, C1, HW, G, eps, act, ws.data_ptr(), ws.numel(), s)
    else:
        rc = lib.pgn_groupnorm_silu_nhwc_cat(x.data_ptr(), C1, x2.data_ptr(), x2.shape[2], y.data_ptr(), dt,
                                             gamma.data_ptr(), beta.data_ptr(), dt, ap, dt, N, HW, G, eps, act,
                                             ws.data_ptr(), ws.numel(), s)
    assert rc == 0, lib.pgn_error_string(rc)


def _inputs(N, HW, C, dtype, seed, with_add):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(N, HW, C, generator=g) * (0.5 + 2 * torch.rand(1, 1, C, generator=g))
         + 2 * torch.randn(1, 1, C, generator=g)).to(dtype)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    beta = (0.2 * torch.randn(C, generator=g)).to(dtype)
    add = (0.5 * torch.randn(N, C, generator=g)).to(dtype) if with_add else None
    return x, gamma, beta, add


def _oracle(x, gamma, beta, add, G, eps, act):
    N, HW, C = x.shape
    xs = x if add is None else x + add[:, None, :]
    nchw = xs.permute(0, 2, 1).reshape(N, C, HW, 1)
    return ops_ref.group_norm_silu_ref(nchw, G, gamma, beta, eps, bool(act)).reshape(N, C, HW).permute(0, 2, 1)


CASES = [
    # N, HW, C, G, add, act, dtype
    (2, 128 * 128, 320, 32, False, 1, torch.bfloat16),   # 86 chunks of 192 rows, last one short; 480-thread workgroups
    (3, 64 * 64, 640, 32, True, 1, torch.bfloat16),
    (2, 32 * 32, 1280, 32, True, 0, torch.float16),
    (2, 32 * 32, 2560, 32, False, 1, torch.bfloat16),    # 320 16-byte chunks per row: one row lane
    (2, 76 * 52, 960, 32, False, 1, torch.bfloat16),     # 1216x832: ragged last chunk
    (5, 77, 64, 32, True, 1, torch.float16),             # 2 channels per group: a lane's 8 channels span 4 groups
    (2, 1, 256, 32, False, 0, torch.bfloat16),           # one row
    (1, 300, 32, 32, False, 1, torch.bfloat16),          # one channel per group
    (2, 1000, 48, 6, True, 1, torch.bfloat16),           # 8 channels per group, 6 groups
]


@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}hw{c[1]}c{c[2]}g{c[3]}{'a' if c[4] else ''}{'s' if c[5] else ''}" for c in CASES])
def test_one_pass_agrees_with_three_pass_safety_net_and_oracle(case):
    N, HW, C, G, with_add, act, dtype = case
    lib = _lib.load()
    x, gamma, beta, add = _inputs(N, HW, C, dtype, C + HW, with_add)
    dx, dg, db = x.to(DEV), gamma.to(DEV), beta.to(DEV)
    da = None if add is None else add.to(DEV)
    nbytes = int(lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, G))
    outs = {}
    for name, env in (("one", {"PGN_GN_NHWC_ONEPASS": "1"}), ("three", {"PGN_GN_NHWC_ONEPASS": "0"}),
                      ("net", {"PGN_GN_NHWC_ONEPASS": "1", "PGN_GN_NHWC_ONEPASS_TIMEOUT": "0"})):
        ws = torch.full((nbytes,), 0xA5, dtype=torch.uint8, device=DEV)       # poisoned: nothing may rely on zeros
        y = torch.full_like(dx, float("nan"))
        with _env(**env):
            _call(lib, dx, y, dg, db, da, G, 1e-5, act, ws)
        torch.cuda.synchronize()
        outs[name] = y.cpu()
    assert torch.equal(outs["one"], outs["net"])                             # same words, same order
    want = _oracle(x, gamma, beta, add, G, 1e-5, act)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    for name in ("one", "three"):
        err = (outs[name].float() - want.float()).abs()
        assert bool((err <= ulp * want.float().abs() + 4 * ulp * 2.0 ** -7 + 1e-6).all()), (name, err.max().item())
    assert (outs["one"] != outs["three"]).float().mean().item() < 0.01       # two summation orders: last-bit noise only


def test_one_pass_over_an_unwritten_concatenation_and_in_place():
    lib = _lib.load()
    N, HW, C1, C2, G = 2, 38 * 26, 1280, 640, 32
    x, gamma, beta, add = _inputs(N, HW, C1 + C2, torch.bfloat16, 11, True)
    d1, d2 = x[:, :, :C1].contiguous().to(DEV), x[:, :, C1:].contiguous().to(DEV)
    dg, db, da = gamma.to(DEV), beta.to(DEV), add.to(DEV)
    nbytes = int(lib.pgn_groupnorm_nhwc_workspace_bytes(N, C1 + C2, HW, G))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    y_cat = torch.empty(N, HW, C1 + C2, dtype=torch.bfloat16, device=DEV)
    y_one = torch.empty_like(y_cat)
    dx = x.to(DEV)
    with _env(PGN_GN_NHWC_ONEPASS="1"):
        _call(lib, d1, y_cat, dg, db, da, G, 1e-5, 1, ws, x2=d2)
        _call(lib, dx, y_one, dg, db, da, G, 1e-5, 1, ws)
    torch.cuda.synchronize()
    assert torch.equal(y_cat, y_one)
    want = _oracle(x, gamma, beta, add, G, 1e-5, 1)
    err = (y_one.cpu().float() - want.float()).abs()
    assert bool((err <= 2.0 ** -7 * want.float().abs() + 2.0 ** -12).all()), err.max().item()
    # y == x: the one-pass form must not be picked (peers read the pivot row after it may have been overwritten);
    # the three-pass form is, and its result is the three-pass result
    with _env(PGN_GN_NHWC_ONEPASS="0"):
        y_three = torch.empty_like(y_cat)
        _call(lib, dx, y_three, dg, db, da, G, 1e-5, 1, ws)
    inplace = dx.clone()
    with _env(PGN_GN_NHWC_ONEPASS="1"):
        _call(lib, inplace, inplace, dg, db, da, G, 1e-5, 1, ws)
    torch.cuda.synchronize()
    assert torch.equal(inplace, y_three)


def test_one_pass_back_to_back_calls_share_one_workspace():
    """Six different inputs (three shapes) through ONE workspace with no synchronisation between the calls: counters and
    words of the previous call are still in the buffer (and in the caches of the CUs that read them) when the next call
    starts.  Every element of every output is compared with a run of its own on a fresh workspace."""
    lib = _lib.load()
    shapes = [(4, 4096, 640, 32), (2, 1024, 1280, 32), (4, 4096, 640, 32), (3, 16384, 320, 32), (2, 1024, 1280, 32),
              (4, 4096, 640, 32)]
    nbytes = max(int(lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, G)) for N, HW, C, G in shapes)
    shared = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    ins, outs = [], []
    for i, (N, HW, C, G) in enumerate(shapes):
        x, gamma, beta, add = _inputs(N, HW, C, torch.bfloat16, 100 + i, i % 2 == 0)
        ins.append([t if t is None else t.to(DEV) for t in (x, gamma, beta, add)] + [G])
    with _env(PGN_GN_NHWC_ONEPASS="1"):
        for rounds in range(3):
            outs = []
            for dx, dg, db, da, G in ins:
                y = torch.full_like(dx, float("nan"))
                _call(lib, dx, y, dg, db, da, G, 1e-5, 1, shared)
                outs.append(y)
            torch.cuda.synchronize()
            for (dx, dg, db, da, G), y in zip(ins, outs):
                fresh = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
                want = torch.empty_like(dx)
                _call(lib, dx, want, dg, db, da, G, 1e-5, 1, fresh)
                torch.cuda.synchronize()
                assert torch.equal(y, want)


def test_one_pass_at_the_bench_shape_is_reproducible_and_close_to_three_pass():
    """UNet batch 32 at 128 x 128 x 320 (2752 workgroups, five and a half rounds of the chip) and 64 x 64 x 640."""
    lib = _lib.load()
    for (N, HW, C, G) in ((32, 16384, 320, 32), (32, 4096, 640, 32)):
        g = torch.Generator(device=DEV).manual_seed(C)
        dx = (torch.randn(N, HW, C, generator=g, device=DEV) * 1.7 + torch.randn(1, 1, C, generator=g, device=DEV)).bfloat16()
        dg = (1 + 0.2 * torch.randn(C, generator=g, device=DEV)).bfloat16()
        db = (0.2 * torch.randn(C, generator=g, device=DEV)).bfloat16()
        nbytes = int(lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, G))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        ys = []
        with _env(PGN_GN_NHWC_ONEPASS="1"):
            for _ in range(4):
                y = torch.empty_like(dx)
                _call(lib, dx, y, dg, db, None, G, 1e-5, 1, ws)
                ys.append(y)
        with _env(PGN_GN_NHWC_ONEPASS="0"):
            y3 = torch.empty_like(dx)
            _call(lib, dx, y3, dg, db, None, G, 1e-5, 1, ws)
        torch.cuda.synchronize()
        for y in ys[1:]:
            assert torch.equal(y, ys[0])
        assert bool(torch.isfinite(ys[0].float()).all())
        diff = (ys[0].float() - y3.float()).abs()
        assert bool((diff <= 2.0 ** -7 * y3.float().abs() + 2.0 ** -12).all()), diff.max().item()
        assert (ys[0] != y3).float().mean().item() < 0.01
        # per-(sample, group) statistics of the output before the affine map would be (0, 1); with it, compare the
        # two forms' per-sample means instead: a stale partial word would move a whole group of a sample
        m1 = ys[0].float().mean(dim=1)
        m3 = y3.float().mean(dim=1)
        assert (m1 - m3).abs().max().item() < 1e-3


def test_one_pass_random_shapes_against_the_oracle():
    """Thirty seeded shapes: every remainder of rows against the chunk size, 1-64 row lanes, channel counts whose
    16-byte chunks do not fill a wavefront, 1-16 channels per group, with and without the per-sample addend."""
    import random
    lib = _lib.load()
    rnd = random.Random(20241)
    with _env(PGN_GN_NHWC_ONEPASS="1"):
        for i in range(30):
            G = rnd.choice([1, 2, 4, 8, 16, 32, 64])
            cpg = rnd.choice([1, 2, 3, 4, 5, 8, 10, 16])
            C = G * cpg
            while C % 8:
                C += G
            if C > 2048:
                C = G * 8
            N = rnd.randint(1, 5)
            HW = rnd.choice([1, 2, 7, 63, 64, 65, 255, 257, 1000, 1024, 1025, 3000, 4097])
            with_add, act = rnd.random() < 0.5, rnd.randint(0, 1)
            dtype = rnd.choice([torch.bfloat16, torch.float16])
            x, gamma, beta, add = _inputs(N, HW, C, dtype, 500 + i, with_add)
            dev = [t if t is None else t.to(DEV) for t in (x, gamma, beta, add)]
            nbytes = int(lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, G))
            assert nbytes > 0, (N, HW, C, G)
            ws = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device=DEV)
            y = torch.full_like(dev[0], float("nan"))
            _call(lib, dev[0], y, dev[1], dev[2], dev[3], G, 1e-5, act, ws)
            torch.cuda.synchronize()
            want = _oracle(x, gamma, beta, add, G, 1e-5, act)
            ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
            err = (y.cpu().float() - want.float()).abs()
            ok = err <= ulp * want.float().abs() + 4 * ulp * 2.0 ** -7 + 1e-6
            assert bool(ok.all()), (i, N, HW, C, G, with_add, act, dtype, err.max().item())


# the one-pass form sums about ONE pivot per (sample, group) - row 0 of the group's first channel - and takes
# var = E[d^2] - E[d]^2; the three-pass form uses per-channel pivots.  An outlier AT the pivot position makes every
# deviation large and the subtraction cancel.  Both forms are picked at run time by tensor size (batch 4 against batch
# 16), so both must hold the same bar on such inputs.  Pivot element at `sigmas` standard deviations from its group.
@pytest.mark.parametrize("sigmas", [30.0, 100.0, 300.0])
@pytest.mark.parametrize("shape", [(2, 64 * 64, 640, 32), (2, 128 * 128, 320, 32), (2, 32 * 32, 1280, 32)],
                         ids=["c640", "c320", "c1280"])
def test_outlier_at_the_pivot_position(shape, sigmas):
    N, HW, C, G = shape
    lib = _lib.load()
    g = torch.Generator().manual_seed(int(sigmas) + C)
    x = torch.randn(N, HW, C, generator=g)
    cpg = C // G
    x[:, 0, ::cpg] = sigmas                       # row 0 of every group's first channel
    x[1, :, cpg] += 0.25 * sigmas                 # and, in sample 1, one whole channel of group 1 with a large mean
    x = x.to(torch.bfloat16)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(torch.bfloat16)
    beta = (0.2 * torch.randn(C, generator=g)).to(torch.bfloat16)
    xd = x.double().reshape(N, HW, G, cpg)                      # fp64 oracle (F.group_norm's definition)
    mu = xd.mean(dim=(1, 3), keepdim=True)
    var = ((xd - mu) ** 2).mean(dim=(1, 3), keepdim=True)
    want = ((xd - mu) / torch.sqrt(var + 1e-5)).reshape(N, HW, C) * gamma.double() + beta.double()
    want = want * torch.sigmoid(want)
    dx, dg, db = x.to(DEV), gamma.to(DEV), beta.to(DEV)
    ws = torch.empty(int(lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, G)), dtype=torch.uint8, device=DEV)
    ulp = 2.0 ** -7
    for name, env in (("one", {"PGN_GN_NHWC_ONEPASS": "1"}), ("three", {"PGN_GN_NHWC_ONEPASS": "0"})):
        y = torch.empty_like(dx)
        with _env(**env):
            _call(lib, dx, y, dg, db, None, G, 1e-5, 1, ws)
        torch.cuda.synchronize()
        err = (y.cpu().double() - want).abs()
        excess = (err - (ulp * want.abs() + 4 * ulp * 2.0 ** -7 + 1e-6)).max().item()
        print(f"{name}-pass, pivot outlier {sigmas} sigma, C={C}: worst excess over the 1-ulp bar {excess:.3e}")
        assert excess <= 0, (name, sigmas, excess)
