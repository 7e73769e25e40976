"""Scripts that A/B kernel geometries load the MEASUREMENT build of the library (make -C pigeon_amd/csrc tuning ->
pigeon_amd/lib/libpigeon_hip_tuning.so: same sources + -DPGN_TUNING_HOOKS); the shipped library has no such hooks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pigeon_amd import _lib  # noqa: E402


def use_tuning_lib():
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libpigeon_hip_tuning.so")
    if not os.path.exists(path):
        raise SystemExit(f"{path} missing: run `make -C pigeon_amd/csrc tuning` first")
    _lib.LIB_PATH = path
    return _lib.load()
