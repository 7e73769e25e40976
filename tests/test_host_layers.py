"""Host-side layer logic that must hold without a GPU: library-only layers keep the plain diffusers order of
operations on CPU tensors, layers built on a HIP kernel refuse CPU tensors."""
import pytest
import torch

from pigeon_amd import attention_processor as ap
from pigeon_amd.layers import ResnetBlock2D, Upsample2D


def test_resnet_block_has_no_cpu_fallback_in_either_form():
    # up blocks hand torch.cat([hidden, skip], dim=1) over as its two halves (layers.ResnetBlock2D.forward); off the
    # GPU the block would concatenate itself, but its GroupNorm is the HIP kernel: the product path fails loudly
    # instead of computing anything on the CPU (the concatenated and the two-source forms are compared on the GPU,
    # tests/test_gpu_models.py::test_host_side_restructurings_agree_with_the_plain_order)
    torch.manual_seed(0)
    blk = ResnetBlock2D(48, 32, temb_channels=16, groups=8).eval()
    h, skip, temb = torch.randn(2, 32, 6, 5), torch.randn(2, 16, 6, 5), torch.randn(2, 16)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            blk(h, temb, skip=skip)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            blk(torch.cat([h, skip], dim=1), temb)


def test_upsample_cpu_path_is_torch_nearest():
    torch.manual_seed(1)
    up = Upsample2D(8).eval()
    x = torch.randn(1, 8, 3, 4)
    with torch.no_grad():
        want = up.conv(torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest"))
        assert torch.equal(up(x), want)
        assert torch.equal(up(x, output_size=(6, 8)), want)
        assert up(x, output_size=(7, 9)).shape == (1, 8, 7, 9)


def test_self_attention_on_cpu_uses_separate_projections_and_keeps_the_state_dict():
    torch.manual_seed(2)
    attn = ap.Attention(32, heads=2, dim_head=16).eval()
    keys = set(attn.state_dict())
    x = torch.randn(2, 5, 32)
    with torch.no_grad():
        q, k, v = ap.HipAttnProcessor._self_qkv(attn, x)
    assert torch.equal(q, attn.to_q(x)) and torch.equal(k, attn.to_k(x)) and torch.equal(v, attn.to_v(x))
    assert not hasattr(attn, "_qkv_fused") and set(attn.state_dict()) == keys


def test_pending_bias_totals_follow_in_place_weight_updates():
    # the fp32 bias totals a BasicTransformerBlock carries into its LayerNorms are cached; load_state_dict
    # copies INTO the existing parameters (same data_ptr), so the cache key must see the version bump
    from pigeon_amd.unet import BasicTransformerBlock
    torch.manual_seed(3)
    blk = BasicTransformerBlock(32, 2, 16, 24).eval().requires_grad_(False)
    b1, b2, b3 = blk.attn1.to_out[0].bias, blk.attn2.to_out[0].bias, blk.ff.net[2].bias
    c1, c12, c123 = blk._pending_biases(None)
    assert torch.equal(c1, b1.float()) and torch.equal(c12, b1.float() + b2.float())
    assert torch.equal(c123, b1.float() + b2.float() + b3.float())
    assert blk._pending_biases(None)[2] is c123                        # cached while nothing changed
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    sd["attn2.to_out.0.bias"] += 1.0
    ptr = b2.data_ptr()
    blk.load_state_dict(sd)
    assert blk.attn2.to_out[0].bias.data_ptr() == ptr                  # in place: storage unchanged
    n1, n12, n123 = blk._pending_biases(None)
    assert torch.equal(n1, c1) and torch.allclose(n12, c12 + 1.0) and torch.allclose(n123, c123 + 1.0)
    # an incoming pending vector (what the previous block of the chain owes) is part of the totals and of the key
    cb = torch.full((32,), 0.25)
    assert torch.allclose(blk._pending_biases(cb)[0], n1 + 0.25)
    cb += 1.0
    assert torch.allclose(blk._pending_biases(cb)[0], n1 + 1.25)


def test_derived_constants_follow_in_place_parameter_updates_and_silu_is_computed_once():
    # the resnets' fp32 conv1 bias, the summed conv2 + shortcut bias and Transformer2DModel's folded projection bias
    # are kept between forwards (layers.derived_constant): same key discipline as the pending-bias totals
    import copy
    from pigeon_amd.layers import derived_constant, silu_once
    torch.manual_seed(4)
    blk = ResnetBlock2D(48, 32, temb_channels=16, groups=8).eval().requires_grad_(False)
    b = blk._shortcut_bias(torch.bfloat16)
    assert torch.equal(b, (blk.conv2.bias.float() + blk.conv_shortcut.bias.float()).to(torch.bfloat16))
    assert blk._shortcut_bias(torch.bfloat16) is b                      # kept
    assert blk._shortcut_bias(torch.float16).dtype == torch.float16     # one slot per dtype
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    sd["conv2.bias"] += 2.0
    blk.load_state_dict(sd)                                             # in place: same storage, new version
    nb = blk._shortcut_bias(torch.bfloat16)
    assert nb is not b and torch.equal(nb, (blk.conv2.bias.float() + blk.conv_shortcut.bias.float()).to(torch.bfloat16))
    assert set(blk.state_dict()) == set(sd)                             # the kept tensors are not state
    twin = copy.deepcopy(blk)                                           # other storage: the copy rebuilds its own
    twin.conv2.bias.add_(1.0)
    assert torch.equal(twin._shortcut_bias(torch.bfloat16),
                       (twin.conv2.bias.float() + twin.conv_shortcut.bias.float()).to(torch.bfloat16))
    calls = []
    src = torch.randn(8)
    first = derived_constant(blk, "_t", (src, None), lambda: calls.append(1) or src * 2)
    assert derived_constant(blk, "_t", (src, None), lambda: calls.append(1) or src * 2) is first and len(calls) == 1
    # silu of the time embedding: once per tensor object and version
    temb = torch.randn(2, 16)
    s = silu_once(temb)
    assert torch.equal(s, torch.nn.functional.silu(temb)) and silu_once(temb) is s
    temb.mul_(2.0)
    assert torch.equal(silu_once(temb), torch.nn.functional.silu(temb))
    # and the promoted add the resnet uses equals the three-launch form it replaces
    t16 = torch.randn(2, 32).bfloat16()
    assert torch.equal(torch.add(t16, blk.conv1.bias.float()), t16.float() + blk.conv1.bias.float())
