"""Doors onto oracle/_ref — the reference's own native code compiled as it is.

TEST INFRASTRUCTURE ONLY.  `available()` is False where oracle/_ref was never
built (no /root/reference and no prebuilt .so in the snapshot).
"""
import ctypes as C
import glob
import importlib.util
import os

import numpy as np

from . import HERE
from .native import lists_to_csr

_REF_DIR = os.path.join(HERE, "_ref")
_LIB = os.path.join(_REF_DIR, "libneurec_ref.so")


def available():
    return os.path.isfile(_LIB)


def _lib():
    lib = C.CDLL(_LIB)
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    lib.ref_cpp_evaluate_matrix.argtypes = [f32p, C.c_int, C.c_int, i64p, i32p, i32p, C.c_int,
                                            C.c_int, C.c_int, f32p]
    lib.ref_arg_top_k_2d.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p]
    return lib


def eval_matrix(scores, truth_lists, metric_ids, top_k, threads=8):
    """The reference's cpp_evaluate_matrix itself (evaluate.h:53-72)."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    rows, cols = scores.shape
    ptr, idx = lists_to_csr(truth_lists)
    mids = np.asarray(metric_ids, dtype=np.int32)
    out = np.zeros((rows, len(mids) * top_k), dtype=np.float32)
    _lib().ref_cpp_evaluate_matrix(scores, cols, rows, ptr, idx, mids, len(mids), top_k, threads,
                                   out)
    return out


def arg_topk(scores, top_k, threads=8):
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    rows, cols = scores.shape
    out = np.zeros((rows, top_k), dtype=np.int32)
    _lib().ref_arg_top_k_2d(scores, cols, rows, top_k, threads, out)
    return out


def random_choice_module():
    """The reference's Cython module util/cython/random_choice.pyx, compiled as is."""
    hits = glob.glob(os.path.join(_REF_DIR, "random_choice*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("random_choice", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _load_ext(name):
    hits = glob.glob(os.path.join(_REF_DIR, name + "*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location(name, hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sampler_module():
    """The reference's data/sampler.py (PairwiseSampler & co.) with its own util/data_iterator.py and
    util/cython/random_choice.pyx underneath — all three compiled unchanged by oracle/Makefile.
    Their `from util import DataIterator` / `from util.cython.random_choice import ...` /
    `from collections import Iterable` lines are satisfied here with the compiled modules (the
    reference's `util` package itself imports TensorFlow).  None where oracle/_ref is incomplete."""
    import collections
    import collections.abc
    import sys
    import types
    rc, di = random_choice_module(), _load_ext("data_iterator")
    if rc is None or di is None or not glob.glob(os.path.join(_REF_DIR, "sampler*.so")):
        return None
    if not hasattr(collections, "Iterable"):            # data/sampler.py:6 predates python 3.10
        collections.Iterable = collections.abc.Iterable
    saved = {k: sys.modules.get(k) for k in ("util", "util.cython", "util.cython.random_choice")}
    util = types.ModuleType("util")
    util.DataIterator = di.DataIterator
    cy = types.ModuleType("util.cython")
    cy.random_choice = rc
    util.cython = cy
    sys.modules.update({"util": util, "util.cython": cy, "util.cython.random_choice": rc})
    try:
        return _load_ext("sampler")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
