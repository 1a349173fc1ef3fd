import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def golden_eval_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "eval_*.npz")))


def truth_lists(ptr, idx):
    return [idx[ptr[r]:ptr[r + 1]].tolist() for r in range(len(ptr) - 1)]


@pytest.fixture(scope="session")
def hostcheck():
    """Host build of neurec_amd/csrc/nr_core.h (tests/hostcheck)."""
    import ctypes as C
    d = os.path.join(ROOT, "tests", "hostcheck")
    so = os.path.join(d, "libhostcheck.so")
    src = os.path.join(d, "hostcheck.cpp")
    core = os.path.join(ROOT, "neurec_amd", "csrc", "nr_core.h")
    if (not os.path.isfile(so)) or os.path.getmtime(so) < max(os.path.getmtime(src),
                                                              os.path.getmtime(core)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-I", os.path.join(ROOT, "neurec_amd", "csrc"), "-o", so, src])
    lib = C.CDLL(so)
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
    lib.hc_partial_sort_copy.argtypes = [f32p, C.c_int, C.c_int, i32p]
    lib.hc_metric.argtypes = [C.c_int, u8p, C.c_int, C.c_int, f32p]
    lib.hc_pack_key.argtypes = [C.c_float, C.c_uint32]
    lib.hc_pack_key.restype = C.c_uint64
    lib.hc_key_score.argtypes = [C.c_uint64]
    lib.hc_key_score.restype = C.c_float
    lib.hc_key_index.argtypes = [C.c_uint64]
    lib.hc_key_index.restype = C.c_uint32
    lib.hc_permute_index.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    lib.hc_permute_index.restype = C.c_uint64
    lib.hc_draw_negative.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, i32p, C.c_int]
    lib.hc_draw_negative.restype = C.c_int32
    lib.hc_nth_allowed.argtypes = [C.c_int32, i32p, C.c_int]
    lib.hc_nth_allowed.restype = C.c_int32
    for f in (lib.hc_softplus, lib.hc_bpr_loss, lib.hc_bpr_dloss):
        f.argtypes = [C.c_float]
        f.restype = C.c_float
    for f in (lib.hc_adam_dense, lib.hc_adam_sparse):
        f.argtypes = [C.c_int, f32p, f32p, f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_float]
    return lib
