"""GPU: the version-keyed tuning artefacts that ship in-tree are the ones the running libraries accept (the MIOpen
find-db name carries MIOpen's build id, the TunableOp table Validator rows for torch / hipBLASLt / rocBLAS); a
mismatch is a loud warning at PigeonPixelDecoder construction and two flags in bench.py's line, not a silent 3-6x."""
import os
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_running_libraries_accept_the_shipped_tables():
    import pigeon_amd
    shipped = os.path.join(ROOT, "pigeon_amd", "tunableop", "gfx950_bf16.csv")
    assert pigeon_amd.gemm_table_mismatches(shipped) == []
    run = pigeon_amd._running_miopen()
    assert run["arch"] == "gfx950" and run["cus"] == 256 and run["build"]
    st = pigeon_amd.miopen_db_status()
    assert st["hit"], st


def test_constructor_sets_both_flags_without_a_warning():
    from pigeon_amd import synthetic
    tiny = dict(block_out_channels=(64, 128, 128), attention_head_dim=(1, 2, 2), transformer_layers_per_block=(1, 1, 2),
                cross_attention_dim=2048, layers_per_block=2, sample_size=16)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pipe = synthetic.build_random_stack("cuda", "bf16", unet_config=tiny, decoder_depth=1, with_vae=False)
    assert pipe.miopen_db_hit is True
    # True for the first decoder of the process; a process whose caller had TunableOp on already reports False
    assert pipe.tuned_gemms in (True, False)
    import pigeon_amd
    assert pigeon_amd._gemm_tables and all(pigeon_amd._gemm_tables.values())
