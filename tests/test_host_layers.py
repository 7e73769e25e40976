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
