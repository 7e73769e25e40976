#!/usr/bin/env python3
"""Generate tests/golden/token_decoder_depth2.npz and vq_decoder_lowres_depth2.npz by RUNNING THE REFERENCE'S OWN MODULE
(/root/reference/Pigeon/models/modeling_decoder.py, HighresVQDecoder and VQDecoder) in the build container.

The reference file is loaded by path; its only missing import is `timm.models.layers` (three names it never uses,
modeling_decoder.py:7), provided here as an empty placeholder module.  Nothing from the reference is copied into the
repository: the fixture holds inputs and outputs only, the weights are regenerated from oracle.weights.
Run:  python scripts/gen_golden.py     (needs /root/reference; not needed on the GPU box)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ops_ref, weights  # noqa: E402

REF = "/root/reference/Pigeon/models/modeling_decoder.py"


def load_reference_module():
    layers = types.ModuleType("timm.models.layers")
    layers.drop_path = layers.to_2tuple = layers.trunc_normal_ = None  # imported at :7, never referenced
    for name in ("timm", "timm.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["timm.models.layers"] = layers
    spec = importlib.util.spec_from_file_location("ref_modeling_decoder", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_reference_module()
    model = ref.HighresVQDecoder(depth=2).eval()
    sd = weights.synth_like(model, seed=123)
    model.load_state_dict(sd)
    codebook = weights.synth_codebook()
    tokens = weights.synth_tokens(3, seed=7)
    tokens.append(torch.cat([torch.arange(32002, 32002 + 300), torch.tensor([32001])])[: tokens[0].numel()])
    emb, nums = ops_ref.token_gather_ref(tokens, codebook, torch.float32)  # lavit_for_pigeon.py:903-918 (fp32 config)
    with torch.no_grad():
        prompt, pooled = model(emb, nums)
    rows = torch.arange(0, 256, 16)
    out = os.path.join(ROOT, "tests", "golden", "token_decoder_depth2.npz")
    np.savez_compressed(
        out,
        tokens=torch.stack(tokens).numpy(), token_nums=nums.numpy(),
        prompt_rows=rows.numpy(), prompt_embeds_rows=prompt[:, rows].numpy().astype(np.float32),
        prompt_sum=prompt.double().sum(dim=(1, 2)).numpy(), prompt_abs_sum=prompt.double().abs().sum(dim=(1, 2)).numpy(),
        pooled=pooled.numpy().astype(np.float32),
        gather_rows=emb[:8].numpy(),
    )
    print("wrote", out, os.path.getsize(out), "bytes; token_nums", nums.tolist(),
          "prompt std", float(prompt.std()), "pooled std", float(pooled.std()))

    # the lowres twin (modeling_decoder.py:180-254), same inputs, its own seeded weights
    low = ref.VQDecoder(depth=2).eval()
    low.load_state_dict(weights.synth_like(low, seed=321))
    with torch.no_grad():
        xrec = low(emb, nums)
    out = os.path.join(ROOT, "tests", "golden", "vq_decoder_lowres_depth2.npz")
    np.savez_compressed(
        out,
        tokens=torch.stack(tokens).numpy(), token_nums=nums.numpy(),
        rows=rows.numpy(), xrec_rows=xrec[:, rows].numpy().astype(np.float32),
        xrec_sum=xrec.double().sum(dim=(1, 2)).numpy(), xrec_abs_sum=xrec.double().abs().sum(dim=(1, 2)).numpy(),
    )
    print("wrote", out, os.path.getsize(out), "bytes; xrec", tuple(xrec.shape), "std", float(xrec.std()))


if __name__ == "__main__":
    main()
