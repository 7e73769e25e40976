"""Deterministic synthetic weights shared by the golden-vector generator, the tests and bench.py.
TEST INFRASTRUCTURE (lives under oracle/ because the fixtures depend on it bit for bit).

Every tensor is drawn from its own CPU generator seeded by (seed, crc32(name)), so values do not depend on module
construction order or on torch's nn.init internals.  There is no network in the build: all weights on the path are
random-init; `pos_embed` / `query_embed` are N(0, 0.02) instead of the zeros a fresh
HighresVQDecoder starts with (modeling_decoder.py:268-270), otherwise the decoder is degenerate.
"""
import zlib
from typing import Dict, Iterable, Tuple

import torch


def tensor_for(name: str, shape: Tuple[int, ...], seed: int = 123) -> torch.Tensor:
    g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("pos_embed", "query_embed"):
        return torch.randn(shape, generator=g) * 0.02
    if leaf == "positional_embedding":
        return torch.randn(shape, generator=g) / shape[-1] ** 0.5
    if leaf == "bias":
        return torch.randn(shape, generator=g) * 0.02
    if leaf == "weight" and len(shape) == 1:        # norm scales
        return 1.0 + 0.05 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (1.0 / max(fan_in, 1)) ** 0.5


def synth_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 123) -> Dict[str, torch.Tensor]:
    return {name: tensor_for(name, tuple(shape), seed) for name, shape in shapes}


def synth_like(module: torch.nn.Module, seed: int = 123) -> Dict[str, torch.Tensor]:
    """State dict with the module's parameter names/shapes filled deterministically."""
    return synth_state_dict(((k, tuple(v.shape)) for k, v in module.state_dict().items()), seed)


def synth_codebook(num_tokens: int = 16384, dim: int = 32, seed: int = 123) -> torch.Tensor:
    """l2norm(randn) as CodebookEmbedding.__init__ does (modeling_visual_tokenzier.py:59-61)."""
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(num_tokens, dim, generator=g), p=2, dim=-1)


def synth_tokens(batch: int, seed: int = 123, min_len: int = 20, max_len: int = 256):
    """What llama.generate hands to stage B (inference.py:268-282): n ~ U{20..256} codebook ids, then eos 32001,
    then pad id 2 up to the batch maximum."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(min_len, max_len + 1, (batch,), generator=g).tolist()
    width = max(lens) + 1
    out = []
    for n in lens:
        ids = torch.randint(32002, 32002 + 16384, (n,), generator=g)
        out.append(torch.cat([ids, torch.tensor([32001]), torch.full((width - n - 1,), 2, dtype=torch.long)]))
    return out
