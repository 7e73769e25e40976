"""CPU suite: the op-level oracle against torch's own ops and closed-form answers."""
import math

import torch
import torch.nn.functional as F

from oracle import ops_ref


def test_gather_oracle_edge_cases():
    cb = F.normalize(torch.randn(16384, 32, generator=torch.Generator().manual_seed(0)), dim=-1)
    toks = [torch.tensor([32002, 2, 32001, 32000, 32002 + 16383]), torch.tensor([2, 2]),
            torch.arange(32002, 32002 + 300)]
    emb, nums = ops_ref.token_gather_ref(toks, cb)
    assert nums.tolist() == [2, 0, 256]
    assert emb.dtype == torch.bfloat16 and emb.shape == (258, 32)
    assert torch.equal(emb[0], cb[0].bfloat16()) and torch.equal(emb[1], cb[16383].bfloat16())
    assert torch.equal(emb[2:], cb[:256].bfloat16())
    pad, mask = ops_ref.pad_and_mask_ref(emb, nums)
    assert pad.shape == (3, 256, 32) and mask.sum().item() == 258
    assert pad[1].abs().sum() == 0 and torch.equal(pad[0, :2], emb[:2])


def test_ddim_oracle_equals_explicit_rounding_and_closed_form():
    g = torch.Generator().manual_seed(1)
    eu, et = torch.randn(4096, generator=g).bfloat16(), torch.randn(4096, generator=g).bfloat16()
    x = torch.randn(4096, generator=g)
    a_t, a_p = torch.tensor(0.31), torch.tensor(0.47)
    for gs in (7.0, 7.1):
        eps = ops_ref.cfg_combine_ref(torch.cat([eu, et]), gs)
        assert eps.dtype == torch.bfloat16
        prev, x0 = ops_ref.ddim_step_ref(eps, x, a_t, a_p)
        prev2, x02 = ops_ref.ddim_step_explicit(eu, et, x, float(torch.tensor(gs, dtype=torch.float32)), a_t ** 0.5,
                                                (1 - a_t) ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5, torch.bfloat16)
        assert torch.equal(prev, prev2) and torch.equal(x0, x02)
    # fp32 closed form: DDIM with eps == true noise recovers x0 exactly and re-noises it to alpha_prev
    x0_true, noise = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    xt = a_t.sqrt() * x0_true + (1 - a_t).sqrt() * noise
    prev, x0 = ops_ref.ddim_step_ref(noise, xt, a_t, a_p)
    assert torch.allclose(x0, x0_true, atol=1e-5)
    assert torch.allclose(prev, a_p.sqrt() * x0_true + (1 - a_p).sqrt() * noise, atol=1e-5)
    # v-prediction and sample-prediction parametrisations of the same state agree
    v = a_t.sqrt() * noise - (1 - a_t).sqrt() * x0_true
    prev_v, _ = ops_ref.ddim_step_ref(v, xt, a_t, a_p, "v_prediction")
    prev_s, _ = ops_ref.ddim_step_ref(x0_true, xt, a_t, a_p, "sample")
    assert torch.allclose(prev_v, prev, atol=1e-5) and torch.allclose(prev_s, prev, atol=1e-5)


def test_euler_oracle_closed_form():
    g = torch.Generator().manual_seed(2)
    x0_true, noise = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    sigma, sigma_next = torch.tensor(4.0), torch.tensor(2.5)
    xt = x0_true + sigma * noise
    prev, x0 = ops_ref.euler_step_ref(noise, xt, sigma, sigma_next)
    assert torch.allclose(x0, x0_true, atol=1e-5)
    assert torch.allclose(prev, x0_true + sigma_next * noise, atol=1e-5)


def test_attention_oracle_equals_sdpa():
    g = torch.Generator().manual_seed(3)
    B, H, Nq, Nk = 2, 3, 70, 45
    q, k, v = (torch.randn(B, n, H * 64, generator=g) for n in (Nq, Nk, Nk))
    bias = torch.where(torch.rand(B, Nk, generator=g) > 0.3, 0.0, -10000.0)
    want = F.scaled_dot_product_attention(q.view(B, Nq, H, 64).transpose(1, 2), k.view(B, Nk, H, 64).transpose(1, 2),
                                          v.view(B, Nk, H, 64).transpose(1, 2), attn_mask=bias[:, None, None, :])
    got = ops_ref.attention_ref(q, k, v, H, None, bias)
    assert torch.allclose(got, want.transpose(1, 2).reshape(B, Nq, -1), atol=2e-6)


def test_groupnorm_oracle_is_fp32_groupnorm_then_silu():
    x = torch.randn(2, 64, 8, 8).bfloat16()
    w, b = torch.randn(64).bfloat16(), torch.randn(64).bfloat16()
    y = ops_ref.group_norm_silu_ref(x, 32, w, b, 1e-5, True)
    xg = x.float().view(2, 32, -1)
    n = ((xg - xg.mean(-1, keepdim=True)) / (xg.var(-1, unbiased=False, keepdim=True) + 1e-5).sqrt()).view(2, 64, 8, 8)
    z = n * w.float()[None, :, None, None] + b.float()[None, :, None, None]
    assert y.dtype == torch.bfloat16
    assert torch.allclose(y.float(), (z * torch.sigmoid(z)).bfloat16().float(), atol=2e-2)
