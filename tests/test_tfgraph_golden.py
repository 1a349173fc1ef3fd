"""oracle.train (the numpy restatement the GPU parity tests lean on) against traces of the
REFERENCE's own model classes — MF.py / LightGCN.py / NGCF.py / MultiVAE.py imported whole and
unchanged and run under oracle/tf_shim.py (tests/golden/make_golden_tfgraph.py wrote the fixtures).

What this pins: the graph (lookups, regularisers, the rows the loss reads, layer combination, which
Adam form TF applies to which variable, the epoch loop) is the reference's code executed; its
derivatives come from torch.autograd, which shares nothing with the hand-derived backward passes in
oracle/train.py.  Bars: fp64 ≤ 1e-12 everywhere; fp32 ≤ 1e-6 on losses (relative) and first-step
gradients (relative to the gradient's scale); fp32 tables after the steps ≤ 1e-5 (Adam moves a
coordinate by ~lr·sign(g): coordinates whose gradient is at rounding level go either way — the fp64
rows show the restatement itself is exact).
"""
import json

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden
from oracle import train as O

WIDTHS = (("f32", np.float32, 1e-6, 1e-5), ("f64", np.float64, 1e-12, 1e-12))


def _batches(g):
    return [tuple(g["batches"][k, j, :g["batch_len"][k]] for j in range(3)) for k in range(len(g["batch_len"]))]


def _rel(a, b):
    return np.max(np.abs(np.asarray(a, np.float64) - b) / np.maximum(np.abs(b), 1e-30))


def _close(got, want, tol):
    scale = max(float(np.abs(want).max()), 1.0)
    err = float(np.abs(np.asarray(got, np.float64).reshape(want.shape) - want).max())
    assert err <= tol * scale, (err, tol * scale)
    return err


@pytest.mark.parametrize("tag", ["reg0", "reg01"])
def test_mf_restatement_equals_the_reference_graph(tag):
    g = load_golden("tfgraph_mf_" + tag)
    h = json.loads(str(g["hyper"]))
    assert g["sparse_update"].all()          # MF.py:57-58: both tables only feed gathers -> TF's sparse Adam
    for w, dt, tol, tol_tab in WIDTHS:
        P, Q = g["P0"].astype(dt), g["Q0"].astype(dt)
        mP, vP, mQ, vQ = (np.zeros_like(x) for x in (P, P, Q, Q))
        adam = O.Adam(h["learning_rate"], dtype=dt)
        losses, first = [], None
        for u, p, n in _batches(g):
            if first is None:
                first = O.mf_loss_and_grads(P, Q, u, p, n, h["reg_mf"])
            losses.append(O.mf_step(P, Q, mP, vP, mQ, vQ, u, p, n, h["reg_mf"], adam))
        assert _rel(losses, g[w + "_loss"]) <= tol
        _close(first[1], g[w + "_dP"], tol)
        _close(first[2], g[w + "_dQ"], tol)
        for got, key in ((P, "_P"), (Q, "_Q"), (mP, "_m_P"), (vP, "_v_P"), (mQ, "_m_Q"), (vQ, "_v_Q")):
            _close(got, g[w + key], tol_tab)
        # MF.predict (MF.py:120-122) at the epoch's evaluation, users = those with train items
        users = np.flatnonzero(np.diff(g["train_indptr"]) > 0)
        _close(P[users] @ Q.T, g[w + "_ratings"], tol_tab)
        # the "[iter 1 : loss : %f" line (MF.py:110)
        logged = float(str(g[w + "_log_line"]).split("loss : ")[1].split(",")[0])
        assert abs(logged - float(np.sum(np.asarray(losses, np.float64)) / len(losses))) <= 2e-6 * logged + 1e-6


def test_mf_loss_and_optimiser_variants_equal_the_reference_graph():
    """util/learner.py:2-41 through MF.train_model(): hinge / square / cross_entropy, gd / adagrad /
    rmsprop / momentum (TF's sparse row updates) — SURVEY §8 f2."""
    g = load_golden("tfgraph_mf_variants")
    cases = json.loads(str(g["cases"]))
    reg, lr = float(g["reg"]), float(g["lr"])
    for ci, (pairwise, loss, learner) in enumerate(cases):
        assert g["c%d_sparse_update" % ci].all()
        for w, dt, tol, tol_tab in WIDTHS:
            P, Q = g["P0"].astype(dt), g["Q0"].astype(dt)
            if learner == "adam":
                ad = O.Adam(lr, dtype=dt)
                mP, vP, mQ, vQ = (np.zeros_like(x) for x in (P, P, Q, Q))
            else:
                oP, oQ = O.RowOptimizer(learner, lr, P.shape, dt), O.RowOptimizer(learner, lr, Q.shape, dt)
            losses = []
            for s in range(5):
                users, items, third = (g["c%d_%s" % (ci, k)][s] for k in ("users", "items", "third"))
                if not pairwise:
                    third = third.astype(dt)
                l, r, dP, dQ = O.mf_general_loss_and_grads(P, Q, users, items, third, reg, pairwise, loss)
                losses.append(float(l) + float(r))
                if learner == "adam":
                    ad.sparse_swept(P, mP, vP, dP)
                    ad.sparse_swept(Q, mQ, vQ, dQ)
                    ad.advance()
                else:
                    oP.apply(P, dP, users)
                    oQ.apply(Q, dQ, np.concatenate([items, third]) if pairwise else items)
            assert _rel(losses, g["c%d_%s_loss" % (ci, w)]) <= max(tol, 2e-6 if w == "f32" else 0), (loss, learner)
            # gd applies TF's per-occurrence scatter_sub, adagrad/rmsprop divide by a root: fp32 rounding
            # differs by an ulp or two per step between any two orderings
            _close(P, g["c%d_%s_P" % (ci, w)], tol_tab)
            _close(Q, g["c%d_%s_Q" % (ci, w)], tol_tab)


def _lightgcn_inputs(g, adj):
    U, I = int(g["n_users"]), int(g["n_items"])
    R = sp.csr_matrix((np.ones(len(g["train_indices"]), np.float32), g["train_indices"], g["train_indptr"]),
                      shape=(U, I))
    coo = R.tocoo()
    return U, I, coo, O.lightgcn_adjacency(coo.row, coo.col, U, I, adj)


@pytest.mark.parametrize("adj", ["pre", "norm"])
def test_lightgcn_restatement_equals_the_reference_graph(adj):
    g = load_golden("tfgraph_lightgcn_" + adj)
    h = json.loads(str(g["hyper"]))
    U, I, coo, A = _lightgcn_inputs(g, adj)
    assert int(g["adj_nnz"]) == A.nnz
    assert not g["sparse_update"].any()      # LightGCN.py:135: the tables feed a concat -> dense ApplyAdam
    for w, dt, tol, tol_tab in WIDTHS:
        A_ = A.astype(dt)
        At = A_.T.tocsr()
        e, m, v = g["E0"].astype(dt), np.zeros(g["E0"].shape, dt), np.zeros(g["E0"].shape, dt)
        adam = O.Adam(h["lr"], dtype=dt)
        losses, first = [], None
        for u, p, n in _batches(g):
            if first is None:
                first = O.lightgcn_loss_and_grad(A_, At, e, U, h["n_layers"], u, p, n, h["reg"])
            losses.append(O.lightgcn_step(A_, At, e, m, v, U, h["n_layers"], u, p, n, h["reg"], adam))
        assert _rel(losses, g[w + "_loss"]) <= tol                # (mf_loss, emb_loss) per step
        _close(first[2], g[w + "_dE"], tol)
        for got, key in ((e, "_E"), (m, "_m"), (v, "_v")):
            _close(got, g[w + key], tol_tab)
        Estar, _ = O.lightgcn_propagate(A_, e, h["n_layers"])      # assign_opt + batch_ratings (:112-119)
        users = sorted(set(coo.row.tolist()))
        _close(Estar[users] @ Estar[U:].T, g[w + "_ratings"], tol_tab)


NGCF_W = ("W_gc", "b_gc", "W_bi", "b_bi")


@pytest.mark.parametrize("tag", ["drop", "reg", "wide_64x3", "wide_24_32_8"])
def test_ngcf_restatement_equals_the_reference_graph(tag):
    """wide_*: embedding_size / layer_size other than the shipped 16 / [16, 16] (NGCF.py:31-33,271-286 take any;
    64 / [64, 64, 64] is the NGCF paper's setting) — the reference class run at those widths."""
    g = load_golden("tfgraph_ngcf_" + tag)
    h = json.loads(str(g["hyper"]))
    L = len(h["layer_size"])
    step_masks = (lambda s: [g["masks_%d" % k][s] for k in range(L)]) if "masks_0" in g else \
        (lambda s: [g["masks"][s, k] for k in range(L)])
    eval_masks = [g["eval_masks_%d" % k] for k in range(L)] if "masks_0" in g else list(g["eval_masks"])
    U, I = int(g["n_users"]), int(g["n_items"])
    R = sp.csr_matrix((g["train_data"], g["train_indices"], g["train_indptr"]), shape=(U, I))
    A = O.ngcf_adjacency(R, "norm")
    keep = 1 - h["mess_dropout_ratio"]
    # W_mlp_* exist (NGCF.py:283-286) but the ngcf graph never reads them: no gradient, never updated
    assert sorted(g["updated"].tolist()) == sorted(["user_embedding", "item_embedding"] +
                                                   ["%s_%d" % (n, k) for n in NGCF_W for k in range(L)])
    assert not g["sparse_update"].any()
    for w, dt, tol, tol_tab in WIDTHS:
        A_ = A.astype(dt)
        At = A_.T.tocsr()
        At.sort_indices()
        e = g["E0"].astype(dt)
        W = [[g["%s_%d_0" % (nm, k)].astype(dt) for nm in NGCF_W] for k in range(L)]
        params = [e] + [x for ws in W for x in ws]
        ms, vs = [np.zeros_like(p) for p in params], [np.zeros_like(p) for p in params]
        adam = O.Adam(h["learning_rate"], dtype=dt)
        losses, first = [], None
        for s, (u, p, n) in enumerate(_batches(g)):
            masks = [m.astype(dt) for m in step_masks(s)]
            loss, dE, wg = O.ngcf_loss_and_grads(A_, At, e, [tuple(ws) for ws in W], masks, keep, U, u, p, n,
                                                 h["reg"])
            if first is None:
                first = (dE, wg)
            for pp, m, v, gg in zip(params, ms, vs, [dE] + [x for gs in wg for x in gs]):
                adam.dense(pp, m, v, gg.reshape(pp.shape))
            adam.advance()
            losses.append(float(loss))
        assert _rel(losses, g[w + "_loss"]) <= tol
        _close(first[0], g[w + "_dE"], tol)
        for k in range(L):
            for j, nm in enumerate(NGCF_W):
                # a weight gradient is a 288-term fp32 reduction: numpy's and torch's summation orders
                # differ by a few ulp of the sum
                _close(first[1][k][j], g["%s_d%s_%d" % (w, nm, k)], 3 * tol)
                # Adam moves a coordinate by ~lr whatever |g|: where fp32 does not resolve the gradient the update is
                # rounding-order dependent — the reference's own fp32-vs-fp64 gap is the bar
                key = "%s_%d" % (nm, k)
                bar = 2 * float(np.abs(g["f32_" + key] - g["f64_" + key]).max()) if w == "f32" else 0.0
                _close(W[k][j], g["%s_%s" % (w, key)], tol_tab + bar)
        bar = 2 * float(np.abs(g["f32_E"] - g["f64_E"]).max()) if w == "f32" else 0.0
        _close(e, g[w + "_E"], tol_tab + bar)
        # evaluate(): a forward pass with fresh dropout masks (always on, NGCF.py:193), then np.matmul
        out, _ = O.ngcf_forward(A_, e, [tuple(ws) for ws in W], [mm.astype(dt) for mm in eval_masks], keep)
        _close(out[:U], g[w + "_eval_user_emb"], tol_tab)
        users = np.flatnonzero(np.diff(g["train_indptr"]) > 0)
        _close(out[users] @ out[U:].T, g[w + "_ratings"], tol_tab)


VAE_NAMES = ("Wq0", "bq0", "Wq1", "bq1", "Wp0", "bp0", "Wp1", "bp1")


def _vae_args(p):
    return [p["Wq0"], p["Wq1"]], [p["bq0"], p["bq1"]], [p["Wp0"], p["Wp1"]], [p["bp0"], p["bp1"]]


@pytest.mark.parametrize("tag", ["tanh", "relu_reg"])
def test_multivae_restatement_equals_the_reference_graph(tag):
    g = load_golden("tfgraph_multivae_" + tag)
    h = json.loads(str(g["hyper"]))
    U, I = int(g["n_users"]), int(g["n_items"])
    R = sp.csr_matrix((np.ones(len(g["train_indices"])), g["train_indices"], g["train_indptr"]), shape=(U, I))
    assert g["rows"].shape == (U // h["batch_size"], h["batch_size"])       # MultiVAE.py:146: the tail is dropped
    for w, dt, tol, tol_tab in WIDTHS:
        p = {k: g[k + "_0"].astype(dt) for k in VAE_NAMES}
        params = [p[k] for k in VAE_NAMES]
        ms, vs = [np.zeros_like(x) for x in params], [np.zeros_like(x) for x in params]
        adam = O.Adam(h["learning_rate"], dtype=dt)
        losses, first = [], None
        for s, rows in enumerate(g["rows"]):
            X = np.asarray(R[rows].todense(), dtype=dt)
            Wq, bq, Wp, bp = _vae_args(p)
            loss, (gWq, gbq, gWp, gbp), _ = O.multivae_loss_and_grads(
                X, Wq, bq, Wp, bp, g["drop_masks"][s].astype(dt), dt(0.8), g["eps"][s].astype(dt),
                g["anneal"][s], h["reg"], h["activation"])
            grads = [gWq[0], gbq[0], gWq[1], gbq[1], gWp[0], gbp[0], gWp[1], gbp[1]]
            if first is None:
                first = grads
            for pp, m, v, gg in zip(params, ms, vs, grads):
                adam.dense(pp, m, v, gg.reshape(pp.shape))
            adam.advance()
            losses.append(float(loss))
        assert _rel(losses, g[w + "_loss"]) <= tol
        for a, k in zip(first, VAE_NAMES):
            _close(a, g["%s_d%s" % (w, k)], tol)
        for a, k in zip(params, VAE_NAMES):
            _close(a, g["%s_%s" % (w, k)], tol_tab)
        # predict() (MultiVAE.py:195-203) never clears its rating row between users
        acc, outs = np.zeros((1, I), dt), []
        for u in g["ratings_users"]:
            acc[0, R[u].indices] = 1
            Wq, bq, Wp, bp = _vae_args(p)
            logits, _, _, _ = O.multivae_forward(acc, Wq, bq, Wp, bp, np.ones_like(acc), 1.0,
                                                 np.zeros((1, 16), dt), 0.0, h["activation"])
            outs.append(logits[0])
        _close(np.asarray(outs), g[w + "_ratings"], tol_tab)
        logged = float(str(g[w + "_log_line"]).split("loss : ")[1].split(",")[0])
        assert abs(logged - float(np.sum(losses)) / U) <= 1e-5            # MultiVAE.py:176: divided by num_users


# ------------------------------------------------------------------ the fixtures are what the reference produces
def test_fixtures_regenerate_from_the_reference_tree(tmp_path, monkeypatch):
    """where /root/reference exists (the build container), run the reference classes again and compare
    with the committed files: the fixtures are not hand-edited and the shim has not drifted"""
    from oracle import ref_models
    if not ref_models.available():
        pytest.skip("reference tree not present")
    import importlib.util
    import os
    from conftest import GOLDEN
    spec = importlib.util.spec_from_file_location("make_golden_tfgraph", os.path.join(GOLDEN, "make_golden_tfgraph.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    monkeypatch.setattr(mk, "HERE", str(tmp_path))
    mk.golden_mf(0.01, "reg01")
    mk.golden_lightgcn("pre")
    mk.golden_ngcf(0.0, 0.1, "drop")
    mk.golden_multivae(0.0, "tanh", "tanh")
    for name in ("tfgraph_mf_reg01", "tfgraph_lightgcn_pre", "tfgraph_ngcf_drop", "tfgraph_multivae_tanh"):
        new = np.load(os.path.join(str(tmp_path), name + ".npz"))
        old = load_golden(name)
        assert sorted(new.files) == sorted(old)
        for k in new.files:
            if k.endswith("_log_line"):
                continue                                     # carries a wall-clock time
            if new[k].dtype.kind == "f":
                # same torch build -> same bits; across builds allow the fp32 rounding of a reduction
                np.testing.assert_allclose(new[k], old[k], rtol=2e-6 if new[k].dtype == np.float32 else 1e-12,
                                           atol=1e-6 if new[k].dtype == np.float32 else 1e-13, err_msg=name + ":" + k)
            else:
                np.testing.assert_array_equal(new[k], old[k], err_msg=name + ":" + k)


def test_config0_epoch_on_real_ml100k_restated():
    """BASELINE configs[0] end to end (tests/golden/make_golden_tfgraph.py:golden_ml100k_epoch — real ml-100k, the
    reference's sampler, MF class, predict() and C++ evaluator): oracle.train.mf_step over the same 157 batches gives
    the same epoch log line and tables, and the oracle evaluator the same metrics from them."""
    from oracle import native
    g = load_golden("tfgraph_ml100k_mf_epoch")
    h = json.loads(str(g["hyper"]))
    U, I = int(g["n_users"]), int(g["n_items"])
    rs = np.random.RandomState(int(g["init_seed"]))
    P = (rs.randn(U, 64) * 0.01).astype(np.float32)
    Q = (rs.randn(I, 64) * 0.01).astype(np.float32)
    st = [np.zeros_like(x) for x in (P, P, Q, Q)]
    adam = O.Adam(h["learning_rate"])
    B, n = h["batch_size"], len(g["users"])
    assert n == 80367 and len(g["f32_loss"]) == 157
    losses = [O.mf_step(P, Q, st[0], st[1], st[2], st[3], g["users"][b:b + B], g["pos"][b:b + B], g["neg"][b:b + B],
                        h["reg_mf"], adam) for b in range(0, n, B)]
    assert _rel(losses, g["f32_loss"]) <= 1e-6 and _rel(losses, g["f64_loss"]) <= 1e-5
    bar = float(g["f32_vs_f64_tables"])
    assert max(np.abs(P - g["f32_P"]).max(), np.abs(Q - g["f32_Q"]).max()) <= 1e-5 + bar
    logged = float(str(g["f32_log_line"]).split("loss : ")[1].split(",")[0])
    assert abs(logged - float(np.sum(np.asarray(losses, np.float64)) / 157)) <= 1e-5 * logged
    users = g["eval_users"]
    S = native.score_gemm(P, users, Q)
    native.mask_train(S, users, g["train_indptr"].astype(np.int64), g["train_indices"])
    truth = [g["test_indices"][g["test_indptr"][u]:g["test_indptr"][u + 1]].tolist() for u in users]
    m = np.mean(native.eval_matrix(S, truth, [1, 2, 4, 3, 5], 20), axis=0)
    # rankings come from tables that differ in the 8th digit: a near-tie may swap (a hit crossing a cut-off moves a
    # metric by 1 / (943 k)); north_star's 1e-5 is met when none does — the observed difference is printed
    print("config 0: NDCG@10 %.8f (reference run %.8f)" % (m[2 * 20 + 9], g["f32_metrics"][2 * 20 + 9]))
    assert np.abs(m - g["f32_metrics"]).max() <= 1e-4


def _wide_params(g, dt):
    n = int(g["n_layers"])
    return n, ([g["Wq%d_0" % i].astype(dt) for i in range(n)], [g["bq%d_0" % i].astype(dt) for i in range(n)],
               [g["Wp%d_0" % i].astype(dt) for i in range(n)], [g["bp%d_0" % i].astype(dt) for i in range(n)])


@pytest.mark.parametrize("tag", ["24x40", "20", "8x16x24"])
def test_multivae_any_p_dim_restatement_equals_the_reference_graph(tag):
    """conf/MultiVAE.properties:3 lists other p_dim ([200, 600], [200]); MultiVAE.py builds len(p_dim) layers each
    way.  oracle.train.multivae_general against the reference class run at two, one and three layers."""
    g = load_golden("tfgraph_multivae_wide_" + tag)
    h = json.loads(str(g["hyper"]))
    U, I = int(g["n_users"]), int(g["n_items"])
    R = sp.csr_matrix((np.ones(len(g["train_indices"])), g["train_indices"], g["train_indptr"]), shape=(U, I))
    for w, dt, tol, tol_tab in WIDTHS:
        n, (Wq, bq, Wp, bp) = _wide_params(g, dt)
        params = Wq + bq + Wp + bp
        ms, vs = [np.zeros_like(x) for x in params], [np.zeros_like(x) for x in params]
        adam = O.Adam(h["learning_rate"], dtype=dt)
        losses, first = [], None
        for s, rows in enumerate(g["rows"]):
            X = np.asarray(R[rows].todense(), dtype=dt)
            loss, (gWq, gbq, gWp, gbp), _, _ = O.multivae_general(
                X, Wq, bq, Wp, bp, g["drop_masks"][s].astype(dt), dt(0.8), g["eps"][s].astype(dt), g["anneal"][s],
                h["reg"], h["activation"])
            grads = gWq + gbq + gWp + gbp
            if first is None:
                first = grads
            for pp, m, v, gg in zip(params, ms, vs, grads):
                adam.dense(pp, m, v, gg.reshape(pp.shape))
            adam.advance()
            losses.append(float(loss))
        assert _rel(losses, g[w + "_loss"]) <= tol
        names = ["Wq%d" % i for i in range(n)] + ["bq%d" % i for i in range(n)] + \
                ["Wp%d" % i for i in range(n)] + ["bp%d" % i for i in range(n)]
        for a, k in zip(first, names):
            _close(a, g["%s_d%s" % (w, k)], 3 * tol)
        for a, k in zip(params, names):
            _close(a, g["%s_%s" % (w, k)], tol_tab)
        acc, outs = np.zeros((1, I), dt), []
        for u in g["ratings_users"]:                                   # predict(): the accumulating row
            acc[0, R[u].indices] = 1
            _, _, _, logits = O.multivae_general(acc, Wq, bq, Wp, bp, np.ones_like(acc), 1.0,
                                                 np.zeros((1, Wp[0].shape[0]), dt), 0.0, h["reg"], h["activation"],
                                                 is_training=0.0, want_grads=False)
            outs.append(logits[0])
        _close(np.asarray(outs), g[w + "_ratings"], tol_tab)
