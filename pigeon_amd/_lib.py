"""ctypes binding of libpigeon_hip.so (C ABI declared in include/pigeon_hip.h).

The library is built in-tree (pigeon_amd/lib/libpigeon_hip.so) by `make -C pigeon_amd/csrc` or
`__graft_entry__.build()`.  There is NO fallback: a missing library is an ImportError at first use.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpigeon_hip.so")

PGN_F32, PGN_BF16, PGN_F16 = 0, 1, 2
PGN_PRED_EPSILON, PGN_PRED_SAMPLE, PGN_PRED_V = 0, 1, 2
PGN_STEP_FUSED_FP32, PGN_STEP_CLIP_SAMPLE, PGN_STEP_USE_CLIPPED_OUTPUT = 1, 2, 4
PGN_OK, PGN_ERR_INVALID, PGN_ERR_UNSUPPORTED, PGN_ERR_SCRATCH = 0, -1, -2, -3
PGN_GN_STATIC_SCRATCH_BYTES = 16 << 20

_c = ctypes
_vp, _i, _i64, _f, _u = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_uint

# name -> (restype, argtypes); mirrors include/pigeon_hip.h one to one
SIGNATURES = {
    "pgn_version": (_i, []),
    "pgn_error_string": (_c.c_char_p, [_i]),
    "pgn_workspace_bytes": (_c.c_size_t, [_i]),
    "pgn_token_gather": (_i, [_vp, _vp, _i, _vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "pgn_token_embed": (_i, [_vp, _vp, _i, _vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "pgn_cfg_ddim_step": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i64, _f, _f, _f, _f, _f, _i, _u, _f, _vp]),
    "pgn_cfg_euler_step": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i64, _f, _f, _f, _i, _u, _vp]),
    "pgn_cfg_expand_scale": (_i, [_vp, _i, _vp, _i, _i64, _f, _vp]),
    "pgn_groupnorm_silu": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "pgn_groupnorm_workspace_bytes": (_c.c_size_t, [_i, _i, _i, _i]),
    "pgn_groupnorm_silu_ex": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _c.c_size_t,
                                   _vp]),
    "pgn_groupnorm_nhwc_workspace_bytes": (_c.c_size_t, [_i, _i, _i, _i]),
    "pgn_groupnorm_silu_nhwc": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _c.c_size_t, _vp]),
    "pgn_groupnorm_silu_nhwc_static": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "pgn_groupnorm_silu_nhwc_cat": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _i, _vp,
                                         _c.c_size_t, _vp]),
    "pgn_upsample_nearest2x_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pgn_softmax_rows": (_i, [_vp, _i64, _vp, _i, _i64, _i64, _i64, _f, _vp]),
    "pgn_geglu": (_i, [_vp, _vp, _i, _i64, _i, _i64, _i64, _vp]),
    "pgn_clip_scores": (_i, [_vp, _i, _i64, _vp, _i, _i64, _vp, _i64, _i, _vp, _vp, _i64, _i, _vp]),
    "pgn_bias_residual_add": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i64, _i, _i64, _vp]),
    "pgn_bias_residual_add_stats_bytes": (_c.c_size_t, [_i, _i, _i]),
    "pgn_bias_residual_add_stats": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _c.c_size_t, _vp]),
    "pgn_groupnorm_silu_nhwc_pre": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp,
                                         _c.c_size_t, _vp]),
    "pgn_add_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _f, _vp]),
    "pgn_add_layernorm_cb": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i64, _i, _f, _vp]),
    "pgn_image_to_uint8": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "pgn_attention_kernel_class": (_i, [_i, _i, _i, _i, _i, _i]),
    "pgn_attention_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i] + [_i64] * 12 + [_vp, _i64, _f, _vp]),
    "pgn_attention_causal_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [_i64] * 12 + [_f, _vp]),
}

_lib = None


class PigeonHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises ImportError when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads the process-wide HIP runtime first so both share one libamdhip64)

    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C pigeon_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). pigeon_amd has no fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code == 0:
        return
    msg = load().pgn_error_string(int(code)).decode()
    if code == PGN_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    if code == PGN_ERR_SCRATCH:
        raise MemoryError(f"{what}: {msg}")
    if code == PGN_ERR_INVALID:
        raise ValueError(f"{what}: {msg}")
    raise PigeonHipError(f"{what}: HIP error {code}: {msg}")
