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


def _csv_validators(path):
    """The `Validator,NAME,VALUE` rows of a TunableOp table: the library builds it was tuned against."""
    out = {}
    with open(path) as fh:
        for ln in fh:
            parts = ln.rstrip("\n").split(",", 2)
            if len(parts) == 3 and parts[0] == "Validator":
                out[parts[1]] = parts[2]
    return out


def gemm_table_mismatches(path, running=None):
    """Validator rows of the table at `path` that differ from the running libraries ({name: value}, default: what
    torch's TunableOp reports on this GPU).  TunableOp refuses the WHOLE table on any mismatch (read_file -> False) and
    every Linear then runs the library heuristic: ['NAME: table X, running Y', ...] says which row did it."""
    if running is None:
        import torch.cuda.tunable as tn
        running = dict(tn.get_validators())
    table = _csv_validators(path)
    return [f"{k}: table {v!r}, running {running.get(k)!r}" for k, v in table.items() if running.get(k) != v]


_gemm_tables = {}     # path -> loaded?  (a second PigeonPixelDecoder in the process reuses the first one's answer)


def enable_tuned_gemms(path=None, _tn=None):
    """Turn torch's TunableOp on in LOOK-UP mode with the GEMM table shipped in-tree (pigeon_amd/tunableop/, produced
    by scripts/tune_gemm.py): Linear layers whose (M, N, K) are in the table run the measured-fastest hipBLASLt /
    rocBLAS solution, every other shape keeps the library heuristic; nothing is tuned at run time.  Returns True when
    the table was loaded, and WARNS when it was not: the table is keyed on the torch / hipBLASLt / rocBLAS builds it was
    tuned with (its Validator rows) and is dropped whole on any other build (library kernels).
    PIGEON_TUNABLEOP=0 disables it (no warning: the caller asked)."""
    if _os.environ.get("PIGEON_TUNABLEOP", "1") == "0":
        return False
    path = path or _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tunableop", "gfx950_bf16.csv")
    if path in _gemm_tables:
        return _gemm_tables[path]
    import warnings
    if not _os.path.exists(path):
        warnings.warn(f"pigeon_amd: no GEMM table at {path}; Linear layers run the hipBLASLt heuristic "
                      "(scripts/tune_gemm.py writes one)", stacklevel=3)
        return False
    if _tn is None:
        import torch.cuda.tunable as _tn
    if _tn.is_enabled():          # the caller manages TunableOp itself
        return False
    _tn.enable(True)
    _tn.tuning_enable(False)
    _tn.record_untuned_enable(False)
    ok = bool(_tn.read_file(path))
    if not ok:
        try:
            why = gemm_table_mismatches(path, dict(_tn.get_validators()))
        except Exception as e:   # noqa: BLE001 - the warning must not depend on the diagnosis
            why = [f"(validators unavailable: {e})"]
        warnings.warn(f"pigeon_amd: torch's TunableOp refused the shipped GEMM table {path}: "
                      f"{'; '.join(why) or 'no Validator row differs (file unreadable?)'}.  Every Linear layer of the "
                      "UNet now runs the hipBLASLt heuristic (UNet forward ~5 % slower at batch 16, more at small "
                      "batches); re-tune on this software stack with `python scripts/tune_gemm.py --unet-only`.",
                      stacklevel=3)
    _gemm_tables[path] = ok
    return ok


def _running_miopen():
    """What MIOpen's user find-db file name is made of on this process' GPU 0: arch, CU count, MIOpen version, build id.
    The build id is not exposed by MIOpen's API; torch's wheel builds MIOpen and hipBLASLt from one rocm-libraries
    commit, whose id hipBLASLt's TunableOp validator carries, so it is read from there (None if that fails: the check
    then matches on arch / CUs / version only)."""
    import torch
    prop = torch.cuda.get_device_properties(0)
    v = int(torch.backends.cudnn.version() or 0)
    tweak = None
    try:
        import torch.cuda.tunable as tn
        hv = dict(tn.get_validators()).get("HIPBLASLT_VERSION", "")
        tweak = hv.split("-", 1)[1] if "-" in hv else None
    except Exception:   # noqa: BLE001
        pass
    return {"arch": prop.gcnArchName.split(":")[0], "cus": int(prop.multi_processor_count),
            "version": (v // 1000000, (v // 1000) % 1000, v % 1000), "build": tweak}


def miopen_db_status(running=None, db_dir=None):
    """Does a find-db of the RUNNING MIOpen build exist under MIOPEN_USER_DB_PATH?  MIOpen names the file
    `<arch><CUs in hex>.HIP.<major>_<minor>_<patch>_<build id>.ufdb.txt` and silently ignores every other file, after
    which the immediate-mode heuristic picks 3-6x slower solvers for the SDXL 3x3 convolutions (82 % of a batch-4 run,
    profiles/r01_batch4_untuned_kernel_stats_top.txt).  -> {"hit", "expected", "found"}."""
    import fnmatch
    running = running or _running_miopen()
    db_dir = db_dir or _os.environ.get("MIOPEN_USER_DB_PATH", _db)
    maj, mnr, pat = running["version"]
    expected = f"{running['arch']}{running['cus']:x}.HIP.{maj}_{mnr}_{pat}_{running.get('build') or '*'}.ufdb.txt"
    try:
        found = sorted(f for f in _os.listdir(db_dir) if f.endswith(".ufdb.txt"))
    except OSError:
        found = []
    return {"hit": any(fnmatch.fnmatchcase(f, expected) for f in found), "expected": expected, "found": found,
            "dir": db_dir}


_miopen_warned = False


def warn_if_miopen_db_missed(running=None, db_dir=None):
    """Called at PigeonPixelDecoder construction; warns once per process.  Returns the `hit` flag."""
    global _miopen_warned
    st = miopen_db_status(running, db_dir)
    if not st["hit"] and not _miopen_warned:
        _miopen_warned = True
        import warnings
        warnings.warn(f"pigeon_amd: no MIOpen find-db for this MIOpen build under {st['dir']} (expected "
                      f"{st['expected']}, found {st['found'] or 'nothing'}): MIOpen ignores the shipped db and its "
                      "immediate-mode heuristic picks solvers that run the UNet's 3x3 convolutions several times "
                      "slower.  Run `python scripts/tune_miopen.py` once on this software stack.", stacklevel=3)
    return st["hit"]
