"""NGCF at widths other than the shipped 16 / [16, 16] (neurec_amd/ngcf_wide.py; csrc/ngcf_wide.hip + gemm.hip)
against the reference's own NGCF class run at 64 / [64, 64, 64] (the NGCF paper's setting) and 24 / [32, 8] under
oracle/tf_shim.py (tests/golden/tfgraph_ngcf_wide_*.npz), 1e-5; and the plugin at those widths end to end."""
import json
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-5
NGCF_W = ("W_gc", "b_gc", "W_bi", "b_bi")


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _err(got, want):
    return float(np.abs(np.asarray(got, np.float64).reshape(want.shape) - want).max())


def _rel(got, want):
    return float(np.max(np.abs(np.asarray(got, np.float64) - want) / np.abs(want)))


def _batches(g):
    return [tuple(np.ascontiguousarray(g["batches"][k, j, :g["batch_len"][k]]) for j in range(3))
            for k in range(len(g["batch_len"]))]


@pytest.mark.parametrize("tag", ["64x3", "24_32_8"])
def test_wide_ngcf_engine_equals_the_reference_graph(tag):
    import torch
    from neurec_amd.graph import ngcf_adjacency, transpose_csr
    from neurec_amd.ngcf_wide import NGCFWideEngine
    g = load_golden("tfgraph_ngcf_wide_" + tag)
    h = json.loads(str(g["hyper"]))
    L = len(h["layer_size"])
    U, I = int(g["n_users"]), int(g["n_items"])
    R = sp.csr_matrix((g["train_data"], g["train_indices"], g["train_indptr"]), shape=(U, I))
    A = ngcf_adjacency(R, "norm")
    W0 = [tuple(g["%s_%d_0" % (nm, k)] for nm in NGCF_W) for k in range(L)]
    eng = NGCFWideEngine(A, transpose_csr(A), U, I, g["E0"], W0, h["learning_rate"], h["reg"],
                         h["mess_dropout_ratio"], 128)
    loss2 = torch.zeros(2, device="cuda")
    got, first = [], None
    for s, (u, p, n) in enumerate(_batches(g)):
        eng.step(_dev(u), _dev(p), _dev(n), loss2, masks=[_dev(g["masks_%d" % k][s]) for k in range(L)])
        got.append(float(loss2.cpu().numpy().astype(np.float64).sum()))
        if first is None:                                  # the gradients Adam consumed in the first step
            first = [eng.gE0[:, :eng.w[0]].cpu().numpy().copy()] + \
                    [x.cpu().numpy().copy() for k in range(L) for x in eng.gW[k]]
    assert _rel(got, g["f32_loss"]) <= TOL and _rel(got, g["f64_loss"]) <= TOL
    names = ["%s_%d" % (nm, k) for k in range(L) for nm in NGCF_W]
    dgrad = max([_err(first[0], g["f64_dE"])] + [_err(a, g["f64_d" + nm]) for a, nm in zip(first[1:], names)])
    bar = max([_err(g["f32_E"], g["f64_E"])] + [_err(g["f32_" + nm], g["f64_" + nm]) for nm in names])
    d = [_err(eng.E0.cpu().numpy(), g["f64_E"])]
    for k in range(L):
        for j, nm in enumerate(NGCF_W):
            d.append(_err(eng.W[k][j].cpu().numpy(), g["f64_%s_%d" % (nm, k)]))
    out = eng.forward([_dev(g["eval_masks_%d" % k]) for k in range(L)]).cpu().numpy()
    de = _err(out[:U], g["f64_eval_user_emb"])
    users = np.flatnonzero(np.diff(g["train_indptr"]) > 0)
    dr = _err(out[users].astype(np.float64) @ out[U:].astype(np.float64).T, g["f64_ratings"])
    print("wide NGCF %s: first-step gradients %.1e, parameters after %d steps %.1e (reference fp32-vs-fp64 %.1e), "
          "evaluation embeddings %.1e, ratings %.1e" % (tag, dgrad, len(got), max(d), bar, de, dr))
    assert dgrad <= TOL and max(d) <= TOL + 2 * bar and de <= TOL + 2 * bar and dr <= TOL + 2 * bar


def test_score_gemm_wide_is_the_k_ascending_chain():
    """factor tables of 256 columns (NGCF 64 / [64, 64, 64]) through the general GEMM: the same fmaf chain per score"""
    import torch
    from neurec_amd import engine as E
    from oracle import native
    rng = np.random.RandomState(3)
    P = (rng.randn(300, 256) * 0.2).astype(np.float32)
    Q = (rng.randn(1000, 256) * 0.2).astype(np.float32)
    users = rng.permutation(300)[:170].astype(np.int32)
    gemm = E.score_gemm_for(_dev(Q), 200)
    assert isinstance(gemm, E.ScoreGemmWide)
    got = gemm(_dev(P), _dev(users)).cpu().numpy()[:, :1000]
    np.testing.assert_array_equal(got, native.score_gemm(P, users, Q))
    got = gemm(_dev(P[:150]), None).cpu().numpy()[:, :1000]
    np.testing.assert_array_equal(got, native.score_gemm(P[:150], None, Q))


def test_native_wide_step_is_the_python_launch_sequence_bit_for_bit():
    """nrhip_ngcf_wide_step / _forward (one native call each) against step_reference / forward_reference (the same
    launches issued from Python): identical tables, weights, losses and device-drawn dropout masks."""
    import torch
    from neurec_amd.graph import ngcf_adjacency, transpose_csr
    from neurec_amd.ngcf_wide import NGCFWideEngine
    g = load_golden("tfgraph_ngcf_wide_24_32_8")
    h = json.loads(str(g["hyper"]))
    L = len(h["layer_size"])
    U, I = int(g["n_users"]), int(g["n_items"])
    R = sp.csr_matrix((g["train_data"], g["train_indices"], g["train_indptr"]), shape=(U, I))
    A = ngcf_adjacency(R, "norm")
    W0 = [tuple(g["%s_%d_0" % (nm, k)] for nm in NGCF_W) for k in range(L)]
    mk = lambda: NGCFWideEngine(A, transpose_csr(A), U, I, g["E0"], W0, h["learning_rate"], 1e-3, 0.1, 128)
    a, b = mk(), mk()
    assert a._native is not None
    la, lb = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    for s, (u, p, n) in enumerate(_batches(g)):
        masks = [_dev(g["masks_%d" % k][s]) for k in range(L)] if s % 2 == 0 else None      # given / drawn on the device
        a.step(_dev(u), _dev(p), _dev(n), la, masks=masks)
        b.step_reference(_dev(u), _dev(p), _dev(n), lb, masks=masks)
        assert torch.equal(la, lb)
        for k in range(L):
            assert torch.equal(a.mask[k], b.mask[k])
    assert torch.equal(a.E0p, b.E0p) and torch.equal(a.mE, b.mE)
    for k in range(L):
        for j in range(4):
            assert torch.equal(a.W[k][j], b.W[k][j])
    assert torch.equal(a.forward(), b.forward_reference())
