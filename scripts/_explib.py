"""Loads libneurec_exp.so (csrc/experiments/: the micro-benchmarks behind DESIGN.md's decisions).
Built on demand by `python -m neurec_amd.build --experiments`; not part of the product library."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import _lib, build  # noqa: E402  (the product library must be loaded first)


def load():
    if not os.path.isfile(build.EXP_LIB_PATH):
        build.build_experiments()
    return ctypes.CDLL(build.EXP_LIB_PATH)
