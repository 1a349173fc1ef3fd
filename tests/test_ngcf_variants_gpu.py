"""NGCF's conf surface beyond the shipped defaults, each against the reference class itself
(model/general_recommender/NGCF.py executed under oracle/tf_shim.py; tests/golden/make_golden_tfgraph.py
`golden_ngcf_variants` -> tfgraph_ngcf_variants.npz, fp32 and an fp64 twin, dropout masks and node-dropout draws
carried as data):
  learner      adagrad / rmsprop / gd / momentum (util/learner.py:2-17: every NGCF trainable feeds a dense op, so TF
               runs its dense Apply* kernels) on BOTH NGCF engines (the fused 16-wide one and the width-generic one);
  alg_type     gcn, gcmc (NGCF.py:204-248);
  node dropout NGCF.py:162-164,334-362 (the same draw for the adjacency and its transpose, every step and the
               evaluation forward).
Bars as in tests/test_tfgraph_gpu.py: losses 1e-5 relative, every parameter 1e-5 + the reference's own fp32-vs-fp64
distance."""
import json

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-5
SLOTS = {"ngcf": ("W_gc", "b_gc", "W_bi", "b_bi"), "gcn": ("W_gc", "b_gc", "W_bi", "b_bi"),
         "gcmc": ("W_gc", "b_gc", "W_mlp", "b_mlp")}


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _err(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max())


def _rel(got, want):
    return float(np.max(np.abs(np.asarray(got, np.float64) - want) / np.abs(want)))


def _case(g, name):
    c = {k[len(name) + 1:]: g[k] for k in g if k.startswith(name + "/")}
    return c, json.loads(str(c["hyper"]))


def _run(g, name, engine_kind):
    import torch
    from neurec_amd.graph import ngcf_adjacency, transpose_csr
    from neurec_amd.ngcf_wide import NGCFWideEngine
    from neurec_amd.trainer import NGCFEngine
    c, h = _case(g, name)
    alg = h["alg_type"]
    U, I = int(g["n_users"]), int(g["n_items"])
    R = sp.csr_matrix((g["train_data"], g["train_indices"], g["train_indptr"]), shape=(U, I))
    A = ngcf_adjacency(R, h["adj_type"])
    W0 = [tuple(g["%s_%d_0" % (nm, k)] for nm in SLOTS[alg]) for k in range(2)]
    node = h["node_dropout_ratio"] if (h["node_dropout_flag"] is True and alg == "ngcf") else 0.0
    args = (A, transpose_csr(A), U, I, g["E0"], W0, h["learning_rate"], h["reg"], h["mess_dropout_ratio"], 128)
    if engine_kind == "fused":
        eng = NGCFEngine(*args, learner=h["learner"])
    else:
        eng = NGCFWideEngine(*args, learner=h["learner"], alg_type=alg, node_dropout=node)
    keep_of = None
    if node:
        # the reference draws over norm_adj's stored entries in COO (row-major) order; the engine's CSR is sorted by
        # (row, column): align the draws once
        a = A.tocsr()
        a.sort_indices()
        order = np.lexsort((c["node_cols"], c["node_rows"]))
        assert np.array_equal(c["node_rows"][order], np.repeat(np.arange(a.shape[0]), np.diff(a.indptr)))
        assert np.array_equal(c["node_cols"][order], a.indices)
        keep_of = lambda draw: _dev(np.ascontiguousarray(draw[order]))
    batches = [tuple(np.ascontiguousarray(g["batches"][k, j, :g["batch_len"][k]]) for j in range(3))
               for k in range(len(g["batch_len"]))]
    loss2 = torch.zeros(2, device="cuda")
    got = []
    for s, (u, p, n) in enumerate(batches):
        kw = {} if keep_of is None else {"node_keep_given": keep_of(c["node_keep"][s])}
        eng.step(_dev(u), _dev(p), _dev(n), loss2, masks=[_dev(c["masks"][s, k]) for k in range(2)], **kw)
        got.append(float(loss2.cpu().numpy().astype(np.float64).sum()))
    assert _rel(got, c["f32_loss"]) <= TOL and _rel(got, c["f64_loss"]) <= TOL, (name, got, c["f32_loss"])
    names = [nm for nm in SLOTS[alg] if ("f64_d%s_0" % nm) in c]              # the weights the loss reaches
    assert names == {"ngcf": list(SLOTS["ngcf"]), "gcn": ["W_gc", "b_gc"], "gcmc": list(SLOTS["gcmc"])}[alg]
    bar = max([_err(c["f32_E"], c["f64_E"])] + [_err(c["f32_%s_%d" % (nm, k)], c["f64_%s_%d" % (nm, k)])
                                                 for nm in names for k in range(2)])
    d = [_err(eng.E0.cpu().numpy(), c["f64_E"])]
    for k in range(2):
        for j, nm in enumerate(SLOTS[alg]):
            want = c["f64_%s_%d" % (nm, k)]
            d.append(_err(eng.W[k][j].cpu().numpy().reshape(want.shape), want))   # untouched weights must not move either
    print("NGCF %s (%s engine): all parameters vs the reference graph (fp64) %.1e (its fp32-vs-fp64 %.1e)"
          % (name, engine_kind, max(d), bar))
    assert max(d) <= TOL + bar, (name, d)
    # evaluate(): one more forward with its own dropout (and node-dropout) draws (NGCF.py:140-141)
    kw = {} if keep_of is None else {"node_keep_given": keep_of(c["eval_node_keep"])}
    out = eng.forward([_dev(m) for m in c["eval_masks"]], **kw).cpu().numpy()
    assert _err(out[:U], c["f64_eval_user_emb"]) <= TOL + bar


@pytest.mark.parametrize("learner", ["adagrad", "rmsprop", "gd", "momentum"])
@pytest.mark.parametrize("engine_kind", ["fused", "wide"])
def test_ngcf_other_learners_equal_the_reference_graph(learner, engine_kind):
    _run(load_golden("tfgraph_ngcf_variants"), learner, engine_kind)


@pytest.mark.parametrize("alg", ["gcn", "gcmc"])
def test_ngcf_alg_types_equal_the_reference_graph(alg):
    _run(load_golden("tfgraph_ngcf_variants"), alg, "wide")


def test_ngcf_node_dropout_equals_the_reference_graph():
    _run(load_golden("tfgraph_ngcf_variants"), "nodedrop", "wide")


def test_node_dropout_draws_its_own_masks_when_none_are_given():
    """device-drawn node dropout: ~keep of the entries survive, the transposed matrix holds the same draw, a step runs"""
    import torch
    from neurec_amd.graph import ngcf_adjacency, transpose_csr
    from neurec_amd.ngcf_wide import NGCFWideEngine
    g = load_golden("tfgraph_ngcf_variants")
    U, I = int(g["n_users"]), int(g["n_items"])
    R = sp.csr_matrix((g["train_data"], g["train_indices"], g["train_indptr"]), shape=(U, I))
    A = ngcf_adjacency(R, "norm")
    W0 = [tuple(g["%s_%d_0" % (nm, k)] for nm in SLOTS["ngcf"]) for k in range(2)]
    eng = NGCFWideEngine(A, transpose_csr(A), U, I, g["E0"], W0, 0.005, 0.01, 0.1, 128, node_dropout=0.3)
    v0 = eng.A.vals[:eng.A.nnz].clone()
    u, p, n = (np.ascontiguousarray(g["batches"][0, j, :g["batch_len"][0]]) for j in range(3))
    eng.step(_dev(u), _dev(p), _dev(n), torch.zeros(2, device="cuda"))
    kept = eng.edge_keep[:eng.A.nnz].cpu().numpy().astype(bool)
    assert abs(kept.mean() - 0.7) < 0.03
    v = eng.A.vals[:eng.A.nnz].cpu().numpy()
    assert (v[~kept] == 0).all() and np.allclose(v[kept], v0.cpu().numpy()[kept] / 0.7, rtol=1e-6)
    dense = sp.csr_matrix((v, eng.A.indices[:eng.A.nnz].cpu().numpy(), eng.A.h_indptr), shape=A.shape)
    vt = eng.At.vals[:eng.At.nnz].cpu().numpy()
    dense_t = sp.csr_matrix((vt, eng.At.indices[:eng.At.nnz].cpu().numpy(), eng.At.h_indptr), shape=A.shape)
    assert abs(dense.T.tocsr() - dense_t).max() == 0
    assert np.isfinite(eng.E0.cpu().numpy()).all()
