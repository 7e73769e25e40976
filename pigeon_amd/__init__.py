"""pigeon_amd — MI355X-native image-generation hot path of YiyanXu/Pigeon.

The package holds only what the path of Pigeon/models/lavit_for_pigeon.py:674-789 needs: the HIP
kernels behind a C ABI (csrc/, include/pigeon_hip.h) and the host-side mirror of the diffusers object
protocol the reference drives (UNet2DConditionModel / AttnProcessor / scheduler.step).
"""
__version__ = "0.1.0"

import os as _os

# MIOpen (3x3 convs of the UNet / VAE, reached through torch).  torch's default immediate mode never benchmarks, but its
# heuristic picks a 300-500 TFLOP/s solver for the SDXL shapes where a find reaches 500-900 (CK grouped-conv xdlops).
# A find costs ~4 minutes per model on a fresh machine, so it is done once (scripts/tune_miopen.py) and its text
# find-db ships in-tree: immediate mode reads it and picks the measured-best solver with no find at run time.  Shapes
# that are not in the db (other batch sizes / resolutions) fall back to the heuristic.
_db = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")


def _private_db_copy(src: str) -> str:
    """One process per GPU: MIOpen guards its user db with lock files next to it, and N ranks opening the same in-tree
    files at once queue on them (a lock that times out silently drops that rank to the heuristic solvers, several
    times slower).  Ranks of a multi-process job therefore read a private copy (a few hundred KB)."""
    import atexit, shutil, tempfile
    dst = tempfile.mkdtemp(prefix=f"pigeon_miopen_db_r{_os.environ.get('RANK', '0')}_")
    shutil.copytree(src, dst, dirs_exist_ok=True)
    atexit.register(shutil.rmtree, dst, ignore_errors=True)
    return dst


if _os.path.isdir(_db) and "MIOPEN_USER_DB_PATH" not in _os.environ:
    try:
        _multi = int(_os.environ.get("WORLD_SIZE", "1") or 1) > 1
    except ValueError:
        _multi = False
    try:
        _os.environ["MIOPEN_USER_DB_PATH"] = _private_db_copy(_db) if _multi else _db
    except OSError:          # no writable temporary directory: share the in-tree files
        _os.environ["MIOPEN_USER_DB_PATH"] = _db
_os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # if a caller turns cudnn.benchmark on: db hit or heuristic, no search
_os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")


def tuned_points():
    """(unet_batch, latent_h, latent_w) points whose convolutions are in the shipped find-db (manifest written by
    scripts/tune_miopen.py next to the db).  Empty when MIOPEN_USER_DB_PATH points somewhere else."""
    import json
    path = _os.path.join(_os.environ.get("MIOPEN_USER_DB_PATH", _db), "tuned_points.json")
    try:
        with open(path) as fh:
            return {tuple(p) for p in json.load(fh)}
    except (OSError, ValueError):
        return set()


_warned = set()


def warn_if_untuned(unet_batch: int, latent_h: int, latent_w: int):
    """MIOpen's immediate-mode heuristic picks a ~7 ms CK instance for the SDXL 3x3 convolutions at shapes that are not
    in the find-db (82 % of a batch-4 run, profiles/r01_batch4_untuned_kernel_stats_top.txt): say so once per shape."""
    key = (int(unet_batch), int(latent_h), int(latent_w))
    if key in _warned:
        return
    _warned.add(key)
    pts = tuned_points()
    if pts and key not in pts:
        import warnings
        warnings.warn(f"pigeon_amd: the convolutions of UNet batch {key[0]} at latent {key[1]}x{key[2]} are not in the "
                      f"shipped MIOpen find-db (tuned: {sorted(pts)}); expect them to run several times slower until "
                      f"`MIOPEN_USER_DB_PATH=<db dir> python scripts/tune_miopen.py --batch {key[0] // 2} --height "
                      f"{key[1] * 8} --width {key[2] * 8}` has been run once on this machine", stacklevel=3)


def enable_tuned_gemms(path=None):
    """Turn torch's TunableOp on in LOOK-UP mode with the GEMM table shipped in-tree (pigeon_amd/tunableop/, produced
    by scripts/tune_gemm.py): Linear layers whose (M, N, K) are in the table run the measured-fastest hipBLASLt /
    rocBLAS solution, every other shape keeps the library heuristic; nothing is tuned at run time.  Returns True when
    the table was loaded.  PIGEON_TUNABLEOP=0 disables it."""
    if _os.environ.get("PIGEON_TUNABLEOP", "1") == "0":
        return False
    path = path or _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tunableop", "gfx950_bf16.csv")
    if not _os.path.exists(path):
        return False
    import torch.cuda.tunable as tn
    if tn.is_enabled():          # the caller manages TunableOp itself
        return False
    tn.enable(True)
    tn.tuning_enable(False)
    tn.record_untuned_enable(False)
    return bool(tn.read_file(path))
