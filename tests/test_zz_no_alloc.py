"""GPU test of the allocation-free entry point pgn_groupnorm_silu_nhwc_static (include/pigeon_hip.h): called through
the C ABI, its output must equal pgn_groupnorm_silu_nhwc's bit for bit (same kernels, the workspace in the library's
static device array instead of caller memory) and sit within 1 bf16 ulp of the fp32 oracle, on two calls in a row, the
second with a smaller input (the array keeps the first call's contents).  Collected last on purpose."""
import pytest
import torch

from oracle import ops_ref
from pigeon_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(lib, static, x, gamma, beta, add, G, eps, act):
    """x: [N, HW, C] bf16 device tensor (channels-last rows).  Returns y of the same shape."""
    N, HW, C = x.shape
    y = torch.empty_like(x)
    s = torch.cuda.current_stream().cuda_stream
    ap = None if add is None else add.data_ptr()
    if static:
        rc = lib.pgn_groupnorm_silu_nhwc_static(x.data_ptr(), y.data_ptr(), _lib.PGN_BF16, gamma.data_ptr(),
                                                beta.data_ptr(), _lib.PGN_BF16, ap, _lib.PGN_BF16, N, C, HW, G, eps,
                                                act, s)
    else:
        nbytes = int(lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, G))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        rc = lib.pgn_groupnorm_silu_nhwc(x.data_ptr(), y.data_ptr(), _lib.PGN_BF16, gamma.data_ptr(), beta.data_ptr(),
                                         _lib.PGN_BF16, ap, _lib.PGN_BF16, N, C, HW, G, eps, act, ws.data_ptr(),
                                         nbytes, s)
    assert rc == 0, lib.pgn_error_string(rc)
    torch.cuda.synchronize()
    return y


def test_static_scratch_groupnorm_equals_the_workspace_entry_point():
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    # first a tensor above 32 MiB (three launches: statistics, finalize, apply exchange [N][S][C] partial sums through
    # the array), then two smaller ones that take the one-launch form (arrival counters + [N][S][G] words in the same
    # array, over whatever the call before left there)
    for (N, HW, C, G, with_add, act) in ((8, 4096, 640, 32, True, 1), (4, 4096, 640, 32, True, 1),
                                         (2, 320, 64, 32, False, 0)):
        x = (torch.randn(N, HW, C, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
        gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(torch.bfloat16)
        beta = (0.1 * torch.randn(C, generator=g)).to(torch.bfloat16)
        add = (0.2 * torch.randn(N, C, generator=g)).to(torch.bfloat16) if with_add else None
        dev = [None if t is None else t.to(DEV) for t in (x, gamma, beta, add)]
        assert 0 < lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, G) <= _lib.PGN_GN_STATIC_SCRATCH_BYTES
        want = _run(lib, False, *dev, G, 1e-5, act)
        got = _run(lib, True, *dev, G, 1e-5, act)
        assert torch.equal(got, want)
        # and against the oracle (fp32 GroupNorm of the bf16-rounded sum, NCHW view of the same rows), <= 1 bf16 ulp
        xs = x if add is None else x + add[:, None, :]                     # torch bf16 add: fp32 add, one rounding
        nchw = xs.permute(0, 2, 1).reshape(N, C, HW, 1)
        ref = ops_ref.group_norm_silu_ref(nchw, G, gamma, beta, 1e-5, bool(act)).reshape(N, C, HW).permute(0, 2, 1)
        err = (got.cpu().float() - ref.float()).abs()
        assert bool((err <= 2.0 ** -7 * ref.float().abs() + 2.0 ** -12).all()), err.max().item()


def test_static_scratch_rejects_shapes_beyond_its_size():
    lib = _lib.load()
    N, C, HW, G = 32, 4096, 4096, 32
    assert lib.pgn_groupnorm_nhwc_workspace_bytes(N, C, HW, G) > _lib.PGN_GN_STATIC_SCRATCH_BYTES
    x = torch.empty(16, dtype=torch.bfloat16, device=DEV)   # never touched: the refusal precedes any launch
    rc = lib.pgn_groupnorm_silu_nhwc_static(x.data_ptr(), x.data_ptr(), _lib.PGN_BF16, None, None, _lib.PGN_BF16, None,
                                            0, N, C, HW, G, 1e-5, 1, None)
    assert rc == _lib.PGN_ERR_SCRATCH
    assert b"scratch" in lib.pgn_error_string(rc)
