"""ctypes front end of liboracle.so (oracle/eval_oracle.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

from . import HERE, build

_PATH = os.path.join(HERE, "liboracle.so")
if not os.path.isfile(_PATH):
    build(ref=False)
_lib = C.CDLL(_PATH)

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")

_lib.oracle_eval_matrix.argtypes = [_f32p, C.c_int64, C.c_int, C.c_int, _i64p, _i32p, _i32p,
                                    C.c_int, C.c_int, C.c_int, _f32p, C.c_void_p]
_lib.oracle_arg_topk.argtypes = [_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _i32p]
_lib.oracle_mask_train.argtypes = [_f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p, _i64p, _i32p]
_lib.oracle_score_gemm.argtypes = [_f32p, C.c_int64, C.c_void_p, C.c_int, _f32p, C.c_int64,
                                   C.c_int, C.c_int, _f32p, C.c_int64, C.c_int]
_lib.oracle_randint_choice.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, _i32p]
_lib.oracle_srand.argtypes = [C.c_uint]

METRIC_IDS = {"Precision": 1, "Recall": 2, "MAP": 3, "NDCG": 4, "MRR": 5}


def lists_to_csr(lists):
    """list of per-row id lists -> (indptr int64, indices int32 ascending)."""
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    for r, l in enumerate(lists):
        ptr[r + 1] = ptr[r] + len(l)
    idx = np.zeros(max(int(ptr[-1]), 1), dtype=np.int32)
    for r, l in enumerate(lists):
        idx[ptr[r]:ptr[r + 1]] = np.sort(np.asarray(list(l), dtype=np.int32))
    return ptr, idx


def eval_matrix(scores, truth_lists, metric_ids, top_k, threads=8, want_topk=False):
    """cpp_evaluate_matrix semantics; returns float32 [rows, len(metric_ids)*top_k]."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    rows, cols = scores.shape
    ptr, idx = lists_to_csr(truth_lists)
    mids = np.asarray(metric_ids, dtype=np.int32)
    out = np.zeros((rows, len(mids) * top_k), dtype=np.float32)
    topk = np.zeros((rows, top_k), dtype=np.int32) if want_topk else None
    rc = _lib.oracle_eval_matrix(scores, cols, rows, cols, ptr, idx, mids, len(mids), top_k,
                                 threads, out, topk.ctypes.data if want_topk else None)
    if rc:
        raise ValueError("oracle_eval_matrix rc=%d" % rc)
    return (out, topk) if want_topk else out


def arg_topk(scores, top_k, threads=8):
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    rows, cols = scores.shape
    out = np.zeros((rows, top_k), dtype=np.int32)
    rc = _lib.oracle_arg_topk(scores, cols, rows, cols, top_k, threads, out)
    if rc:
        raise ValueError("oracle_arg_topk rc=%d" % rc)
    return out


def mask_train(scores, users, tr_indptr, tr_indices):
    """In place: scores[r, train(users[r])] = -inf."""
    assert scores.dtype == np.float32 and scores.flags.c_contiguous
    u = None if users is None else np.ascontiguousarray(users, dtype=np.int32)
    _lib.oracle_mask_train(scores, scores.shape[1], scores.shape[0], scores.shape[1],
                           None if u is None else u.ctypes.data,
                           np.ascontiguousarray(tr_indptr, dtype=np.int64),
                           np.ascontiguousarray(tr_indices, dtype=np.int32))
    return scores


def score_gemm(P, users, Q, threads=8):
    """S[r, i] = fmaf-chain over k of P[users[r], k] * Q[i, k]."""
    P = np.ascontiguousarray(P, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    u = None if users is None else np.ascontiguousarray(users, dtype=np.int32)
    rows = P.shape[0] if u is None else len(u)
    S = np.zeros((rows, Q.shape[0]), dtype=np.float32)
    _lib.oracle_score_gemm(P, P.shape[1], None if u is None else u.ctypes.data, rows, Q,
                           Q.shape[1], Q.shape[0], P.shape[1], S, Q.shape[0], threads)
    return S


def srand(seed=1):
    _lib.oracle_srand(seed)


def randint_choice(high, size=1, replace=True, p=None, exclusion=None):
    """randint_choice of util/cython/random_choice.pyx:20-62 (same glibc stream)."""
    if size <= 0:
        raise ValueError("'size' must be a positive integer.")
    if not isinstance(replace, bool):
        raise TypeError("'replace' must be bool.")
    if p is not None:
        raise NotImplementedError
    ex = None if exclusion is None else np.ascontiguousarray(list(exclusion), dtype=np.int32)
    out = np.zeros(size, dtype=np.int32)
    rc = _lib.oracle_randint_choice(high, size, 1 if replace else 0,
                                    None if ex is None else ex.ctypes.data,
                                    0 if ex is None else len(ex), out)
    if rc == 2:
        raise ValueError("The number of 'exclusion' is greater than 'high'.")
    if rc == 3:
        raise ValueError("There is not enough integers to be sampled.")
    return int(out[0]) if size == 1 else out.tolist()


def batch_randint_choice(high, size, replace=True, p=None, exclusion=None):
    """random_choice.pyx:64-89"""
    if p is not None:
        raise NotImplementedError
    if exclusion is not None and len(size) != len(exclusion):
        raise ValueError("The shape of 'exclusion' is not compatible with the shape of 'size'!")
    return [randint_choice(high, size=size[i], replace=replace,
                           exclusion=None if exclusion is None else exclusion[i])
            for i in range(len(size))]
