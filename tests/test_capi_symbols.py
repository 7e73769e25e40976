"""CPU suite: the C-ABI library loads and exports every symbol include/pigeon_hip.h declares (no compute)."""
import ctypes
import os
import re

from pigeon_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pigeon_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgn_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_entry_points():
    names = _declared()
    for must in ("pgn_token_gather", "pgn_cfg_ddim_step", "pgn_cfg_euler_step", "pgn_groupnorm_silu",
                 "pgn_attention_fwd", "pgn_version", "pgn_workspace_bytes"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build with make -C pigeon_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in pigeon_hip.h but not exported"


def test_library_exports_nothing_but_the_declared_symbols():
    # no undeclared entry points (tuning hooks live in the separate measurement build, `make tuning`)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("pgn_")
                       and ln.split()[-2] in ("T", "t")})
    assert exported == _declared()


def test_library_reads_no_environment():
    """include/pigeon_hip.h: "no process-wide switches (no environment variable is read)".  getenv / secure_getenv
    would show up among the undefined (imported) dynamic symbols."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out, [ln for ln in out.splitlines() if "getenv" in ln]


def test_attention_kernel_class_on_the_sdxl_shape_table():
    """ops.attention labels a launch by asking the library (pgn_attention_kernel_class), not by a copy of its rule:
    the SDXL shapes at batch 16 (UNet batch 32) all take the 64-row ring kernel, incl. the ragged 1216x832 ones; small
    launches, a key bias, short key loops and other head dims take the 32-row kernel."""
    lib = _lib.load()
    ring = lambda B, H, Nq, Nk, D=64, bias=0: lib.pgn_attention_kernel_class(B, H, Nq, Nk, D, bias)
    for B, H, Nq, Nk in ((32, 10, 4096, 4096), (32, 20, 1024, 1024), (16, 10, 4096, 256), (16, 20, 1024, 256),
                         (16, 10, 3952, 3952), (16, 20, 988, 988), (4, 10, 16384, 16384)):
        assert ring(B, H, Nq, Nk) == 1, (B, H, Nq, Nk)
    assert ring(2, 10, 4096, 4096) == 0            # batch 1: 320 workgroups of 256 rows leave the chip half empty
    assert ring(8, 20, 988, 256) == 0              # 640 workgroups = 1.25 rounds: the 128-row workgroups fill better
    assert ring(6, 10, 4096, 4096) == 1            # batch 3: 960 workgroups fill it (the fill rule)
    assert ring(32, 10, 4096, 4096, bias=1) == 0 and ring(32, 20, 1024, 192) == 0 and ring(32, 20, 256, 1024) == 0
    assert ring(32, 16, 1024, 1024, D=88) == 0 and ring(32, 16, 1024, 1024, D=264) == -2 and ring(0, 1, 1, 1) == -1


def test_binding_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_error_strings_need_no_gpu():
    lib = _lib.load()
    assert lib.pgn_version() == 112
    assert lib.pgn_error_string(0) == b"ok"
    assert b"invalid" in lib.pgn_error_string(-1)
    assert lib.pgn_workspace_bytes(3) == 0


def test_argument_validation_happens_before_any_launch():
    lib = _lib.load()
    # null pointers / bad sizes are refused on the host, so this is safe without a GPU
    assert lib.pgn_attention_fwd(None, None, None, None, 1, 1, 1, 1, 1, 64, *([0] * 12), None, 0, 0.125, None) == -1
    assert lib.pgn_groupnorm_silu(None, 1, None, 1, None, None, 1, 1, 32, 16, 32, 1e-5, 1, None) == -1
    assert lib.pgn_cfg_ddim_step(None, None, 1, None, None, None, 0, 16, 7.0, 1, 1, 1, 1, 0, 0, 0.0, None) == -1
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    # head_dim above 256 or not a multiple of 8 -> unsupported, 3-channel groups that do not divide -> invalid
    assert lib.pgn_attention_fwd(p16, p16, p16, p16, 1, 1, 1, 1, 1, 264, *([8] * 12), None, 0, 0.1, None) == -2
    assert lib.pgn_attention_fwd(p16, p16, p16, p16, 1, 1, 1, 1, 1, 44, *([8] * 12), None, 0, 0.1, None) == -2
    assert lib.pgn_groupnorm_silu(p16, 1, p16, 1, None, None, 1, 1, 30, 16, 32, 1e-5, 1, None) == -1
    # workspace query is host-only arithmetic; a too-small workspace is refused before any launch
    assert lib.pgn_groupnorm_workspace_bytes(1, 128, 1 << 20, 32) == 32 * 32 * 8
    assert lib.pgn_groupnorm_workspace_bytes(32, 320, 16384, 32) == 0
    assert lib.pgn_groupnorm_workspace_bytes(0, 128, 16, 32) == 0
    assert lib.pgn_groupnorm_silu_ex(p16, 1, p16, 1, None, None, 1, None, 0, 1, 128, 1 << 20, 32, 1e-6, 1, p16, 64, None) == -1
    assert lib.pgn_groupnorm_silu_ex(p16, 1, p16, 1, None, None, 1, p16, 7, 1, 128, 64, 32, 1e-6, 1, None, 0, None) == -1
    assert lib.pgn_image_to_uint8(None, 1, None, 1, 3, 8, 8, None) == -1
    assert lib.pgn_image_to_uint8(p16, 1, p16, 1, 5, 8, 8, None) == -2
    assert lib.pgn_groupnorm_nhwc_workspace_bytes(32, 320, 16384, 32) > 0
    assert lib.pgn_groupnorm_nhwc_workspace_bytes(1, 36, 16, 4) == 0                       # C % 8 != 0: not covered
    assert lib.pgn_groupnorm_silu_nhwc(p16, p16, 1, None, None, 1, None, 0, 1, 64, 16, 32, 1e-5, 1, None, 0, None) == -1
    assert lib.pgn_groupnorm_silu_nhwc(p16, p16, 1, None, None, 1, None, 0, 1, 64, 16, 32, 1e-5, 1, p16, 8, None) == -1
    assert lib.pgn_groupnorm_silu_nhwc(p16, p16, 0, None, None, 1, None, 0, 1, 64, 16, 32, 1e-5, 1, p16, 1 << 20, None) == -2
    # `act` is a bit set (SiLU | PGN_GN_ONEPASS_* | PGN_GN_PIPELINE_*): unknown bits and NEVER with ALWAYS are refused
    assert lib.pgn_groupnorm_silu_nhwc(p16, p16, 1, None, None, 1, None, 0, 1, 64, 16, 32, 1e-5, 64, p16, 1 << 20, None) == -1
    assert lib.pgn_groupnorm_silu_nhwc(p16, p16, 1, None, None, 1, None, 0, 1, 64, 16, 32, 1e-5, 2 | 4, p16, 1 << 20, None) == -1
    assert lib.pgn_groupnorm_silu_nhwc(p16, p16, 1, None, None, 1, None, 0, 1, 64, 16, 32, 1e-5, 16 | 32, p16, 1 << 20, None) == -1
    # concatenation form: second source mandatory, halves in multiples of 8 channels
    cat = lib.pgn_groupnorm_silu_nhwc_cat
    assert cat(p16, 32, None, 32, p16, 1, None, None, 1, None, 0, 1, 16, 32, 1e-5, 1, p16, 1 << 20, None) == -1
    assert cat(p16, 32, p16, 0, p16, 1, None, None, 1, None, 0, 1, 16, 32, 1e-5, 1, p16, 1 << 20, None) == -1
    assert cat(p16, 36, p16, 28, p16, 1, None, None, 1, None, 0, 1, 16, 32, 1e-5, 1, p16, 1 << 20, None) == -2
    assert cat(p16, 32, p16 + 2, 32, p16, 1, None, None, 1, None, 0, 1, 16, 32, 1e-5, 1, p16, 1 << 20, None) == -1
    assert cat(p16, 32, p16, 32, p16, 1, None, None, 1, None, 0, 1, 16, 32, 1e-5, 1, p16, 8, None) == -1   # workspace
    assert lib.pgn_upsample_nearest2x_nhwc(p16, p16, 1, 1, 2, 2, 8, None) == -1            # in place is refused
    assert lib.pgn_upsample_nearest2x_nhwc(p16, p16 + 64, 1, 1, 2, 2, 12, None) == -2      # C % 8
    assert lib.pgn_upsample_nearest2x_nhwc(p16, p16 + 64, 0, 1, 2, 2, 8, None) == -2       # fp32
    assert lib.pgn_upsample_nearest2x_nhwc(p16, p16 + 64, 1, 0, 2, 2, 8, None) == -1
    # row kernels: nulls -> invalid; D / C not a multiple of 8, fp32 io, C > 4096 -> unsupported
    assert lib.pgn_bias_residual_add(p16, None, p16, 1, p16, 1, 2, 12, 1, None) == -2      # C % 8 with inner 1
    assert lib.pgn_bias_residual_add(p16, None, p16, 1, p16, 1, 2, 8, 9, None) == -2       # inner % 8
    assert lib.pgn_bias_residual_add(p16, None, None, 1, p16, 1, 2, 8, 1, None) == -1
    assert lib.pgn_bias_residual_add(p16, None, p16, 1, p16, 0, 2, 8, 1, None) == -2       # fp32 activations
    assert lib.pgn_softmax_rows(p16, 8, p16, 1, 8, 1, 8, 0.0, None) == -1          # scale must be positive
    assert lib.pgn_softmax_rows(p16, 6, p16, 1, 8, 1, 6, 1.0, None) == -1          # leading dimension % 4
    assert lib.pgn_softmax_rows(p16, 8, p16, 0, 8, 1, 8, 1.0, None) == -2          # fp32 probabilities
    assert lib.pgn_add_layernorm_cb(p16, None, None, p16, None, None, 1, 0, p16 + 4, 1, 8, 1e-5, None) == -1   # col_bias alignment
    assert lib.pgn_geglu(None, None, 1, 4, 8, 16, 8, None) == -1
    assert lib.pgn_geglu(p16, p16, 1, 4, 12, 24, 12, None) == -2
    assert lib.pgn_geglu(p16, p16, 0, 4, 8, 16, 8, None) == -2
    assert lib.pgn_geglu(p16, p16, 1, 4, 8, 8, 8, None) == -1            # leading dimension < 2 D
    assert lib.pgn_add_layernorm(None, None, None, None, None, None, 1, 1, 4, 8, 1e-5, None) == -1
    assert lib.pgn_add_layernorm(p16, p16, None, p16, None, None, 1, 1, 4, 8, 1e-5, None) == -1   # residual w/o sum_out
    assert lib.pgn_add_layernorm(p16, None, None, p16, None, None, 1, 1, 4, 12, 1e-5, None) == -2
    assert lib.pgn_add_layernorm(p16, None, None, p16, None, None, 1, 1, 4, 8192, 1e-5, None) == -2


def test_untuned_shape_warning_names_the_tuning_command(recwarn):
    import pigeon_amd
    pts = pigeon_amd.tuned_points()
    assert (32, 128, 128) in pts                      # BASELINE configs[1]: batch 16 -> UNet batch 32, 1024x1024
    pigeon_amd._warned.clear()
    pigeon_amd.warn_if_untuned(32, 128, 128)
    assert len(recwarn) == 0
    pigeon_amd.warn_if_untuned(6, 24, 40)
    pigeon_amd.warn_if_untuned(6, 24, 40)             # once per shape
    assert len(recwarn) == 1 and "tune_miopen.py --batch 3 --height 192 --width 320" in str(recwarn[0].message)


def test_build_entry_point_checks_the_header_version():
    """__graft_entry__.build() compares the loaded library's version with include/pigeon_hip.h (it once compared with a
    literal that an ABI bump left behind)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "__graft_entry__.py")).read()
    assert "PGN_VERSION" in src and not re.search(r"pgn_version\(\)\s*==\s*\d", src)


def test_ranks_of_a_multi_process_job_read_a_private_copy_of_the_find_db():
    """One process per GPU: every rank would otherwise open (and lock) the same in-tree MIOpen user db."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, pigeon_amd; p = os.environ['MIOPEN_USER_DB_PATH']; "
            "print(p); print(sorted(os.listdir(p))); print(sorted(pigeon_amd.tuned_points()))")
    outs = {}
    for tag, extra in (("single", {}), ("rank1", {"WORLD_SIZE": "2", "RANK": "1"})):
        env = {k: v for k, v in os.environ.items() if k not in ("MIOPEN_USER_DB_PATH", "WORLD_SIZE", "RANK")}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = r.stdout.strip().splitlines()
    in_tree = os.path.join(root, "pigeon_amd", "miopen_db")
    assert outs["single"][0] == in_tree
    assert outs["rank1"][0] != in_tree and "pigeon_miopen_db_r1_" in outs["rank1"][0]
    assert outs["rank1"][1:] == outs["single"][1:]              # same files, same tuned points
    assert not os.path.exists(outs["rank1"][0])                 # removed when the process ended
