"""Build-time invariants of the attention kernels that no run-time test can see.

The ring form of the 64-row kernel (pgn_attn_w64r_kernel) orders its LDS-DMA tiles with COUNTED s_waitcnt vmcnt(4)
waits: "all but the newest tile's four DMA instructions have retired".  That arithmetic holds only while the kernel
issues no other vector-memory instruction inside its loop - a register spill would add scratch loads to the same
counter and the wait would let a tile be read before it has landed (rarely, and only on some shapes).  So the kernel
must compile without scratch; hipcc's resource-usage remarks say whether it does (cross-compiles without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_ring_attention_kernel_has_no_scratch_and_two_waves_per_simd(tmp_path):
    src = os.path.join(ROOT, "pigeon_amd", "csrc", "attention.hip")
    # the flags of pigeon_amd/csrc/Makefile for this file (EXTRA_attention included)
    mk = open(os.path.join(ROOT, "pigeon_amd", "csrc", "Makefile")).read()
    extra = re.search(r"^EXTRA_attention\s*:=\s*(.*)$", mk, re.M).group(1).split()
    assert "-fno-slp-vectorize" in extra     # the softmax arithmetic must stay unpacked (see attention.hip)
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", *extra,
                          "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "attention.o")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", out.stderr)
    seen = 0
    for blk in blocks[1:]:
        name = blk.split()[0]
        if "pgn_attn_w64r_kernel" not in name and "pgn_attn_w64_kernel" not in name:
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
        vgprs = int(re.search(r" VGPRs: (\d+)", blk).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", blk).group(1))
        assert scratch == 0, f"{name}: {scratch} bytes of scratch per lane (spills break the counted vmcnt waits)"
        assert vgprs <= 256 and occ >= 2, (name, vgprs, occ)
    assert seen >= 4                     # bf16 + fp16 instantiations of both 64-row kernels


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_ring_attention_softmax_arithmetic_is_not_packed(tmp_path):
    """On gfx950 plain fp32 VALU instructions issue underneath a running MFMA, packed fp32 ones (v_pk_fma_f32,
    v_pk_add_f32) hold the matrix pipe (scripts/ubench_overlap.hip): the shipped ring kernel's common path must not
    contain them.  v_pk_mul_f32 is allowed: it is the O rescale of the rare raise-the-reference branch."""
    src = os.path.join(ROOT, "pigeon_amd", "csrc", "attention.hip")
    mk = open(os.path.join(ROOT, "pigeon_amd", "csrc", "Makefile")).read()
    extra = re.search(r"^EXTRA_attention\s*:=\s*(.*)$", mk, re.M).group(1).split()
    asm = tmp_path / "attention.s"
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", *extra,
                          "--cuda-device-only", "-S", src, "-o", str(asm)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = asm.read_text()
    m = re.search(r"^(\S*pgn_attn_w64r_kernelILi1E\S*):", text, re.M)
    assert m, "bf16 ring kernel not found in the assembly"
    body = text[m.start():text.index(".Lfunc_end", m.start())]
    assert "v_mfma_f32_32x32x16_bf16" in body
    assert "v_pk_fma_f32" not in body and "v_pk_add_f32" not in body


def test_python_sources_reference_no_undefined_global():
    """A name used but never imported / assigned only fails when its line runs - on the GPU box, in a path the CPU suite
    cannot reach (round 3: `os` in pigeon_amd/ops.py).  scripts/lint_undefined.py is a conservative AST check."""
    import glob
    import sys
    files = sorted(glob.glob(os.path.join(ROOT, "pigeon_amd", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "*.py")) +
                   glob.glob(os.path.join(ROOT, "tests", "*.py")) + [os.path.join(ROOT, "bench.py"),
                                                                      os.path.join(ROOT, "__graft_entry__.py")])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lint_undefined.py"), *files],
                         capture_output=True, text=True, timeout=120).stdout
    bad = [ln for ln in out.splitlines() if "__file__" not in ln and "__name__" not in ln]
    assert not bad, bad
