"""The CPU oracle pinned against the reference: golden vectors generated from the
reference's own compiled evaluator/sampler (tests/golden/make_golden.py), the
known-answer vectors of SURVEY.md, and — when oracle/_ref is present — the
reference code itself on fresh random inputs."""
import numpy as np
import pytest

from conftest import golden_eval_cases, load_golden, truth_lists
from oracle import native, ref, train


@pytest.mark.parametrize("case", golden_eval_cases())
def test_eval_oracle_matches_reference_golden(case):
    g = load_golden(case)
    truth = truth_lists(g["truth_ptr"], g["truth_idx"])
    got = native.eval_matrix(g["scores"], truth, g["metrics"].tolist(), int(g["top_k"]))
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, g["result"])          # bit-exact, ties included
    np.testing.assert_array_equal(native.arg_topk(g["scores"], int(g["top_k"])), g["arg_topk"])


def test_known_answer_vector_survey_4():
    g = load_golden("eval_kat")
    r0 = g["result"][0].reshape(5, 4)
    np.testing.assert_allclose(r0[0], [1, .5, .6666667, .5], rtol=0, atol=1e-7)
    np.testing.assert_allclose(r0[1], [.33333334, .33333334, .6666667, .6666667], atol=1e-7)
    np.testing.assert_allclose(r0[2], [1, .5, .5555556, .5555556], atol=1e-7)
    np.testing.assert_allclose(r0[3], [1, .6131472, .7039181, .7039181], atol=1e-7)
    np.testing.assert_allclose(r0[4], [1, 1, 1, 1], atol=0)
    assert not g["result"][1].any()
    assert g["arg_topk"][0].tolist() == [1, 3, 5, 7]


def test_sampler_oracle_replays_reference_glibc_stream():
    g = load_golden("sampler_glibc")
    native.srand(1)
    assert native.randint_choice(1000, size=6, replace=True, exclusion=[1, 2, 3]) == g["first"].tolist()
    assert g["first"].tolist() == [193, 869, 835, 0, 934, 983]
    excl = truth_lists(g["excl_ptr"], g["excl_idx"])
    batch = native.batch_randint_choice(50, g["sizes"].tolist(), replace=True, exclusion=excl)
    flat = np.concatenate([np.atleast_1d(np.asarray(b, np.int32)) for b in batch])
    np.testing.assert_array_equal(flat, g["batch"])
    norep = native.randint_choice(40, size=20, replace=False, exclusion=[0, 1, 2, 3, 4])
    np.testing.assert_array_equal(np.asarray(norep, np.int32), g["norep"])
    assert len(set(norep)) == 20 and not set(norep) & {0, 1, 2, 3, 4}


def test_sampler_oracle_argument_errors():
    with pytest.raises(ValueError):
        native.randint_choice(10, size=0)
    with pytest.raises(ValueError):
        native.randint_choice(3, size=1, exclusion=[0, 1, 2])
    with pytest.raises(ValueError):
        native.randint_choice(10, size=8, replace=False, exclusion=[0, 1])
    with pytest.raises(TypeError):
        native.randint_choice(10, size=1, replace=1)
    with pytest.raises(ValueError):
        native.batch_randint_choice(10, [1, 2], exclusion=[[1]])


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (no /root/reference)")
def test_oracle_equals_live_reference_on_fresh_inputs():
    rng = np.random.RandomState(7)
    for trial in range(12):
        rows, cols = 40, int(rng.randint(60, 2500))
        k = int(rng.randint(1, min(60, cols // 2)))
        s = rng.randn(rows, cols).astype(np.float32)
        if trial % 3 == 1:
            s = (np.round(s * 3) / 3).astype(np.float32)
        if trial % 3 == 2:
            s[rng.rand(rows, cols) < 0.3] = -np.inf
        truth = [np.sort(rng.choice(cols, rng.randint(1, 30), replace=False)).tolist()
                 for _ in range(rows)]
        np.testing.assert_array_equal(native.eval_matrix(s, truth, [1, 2, 3, 4, 5], k),
                                      ref.eval_matrix(s, truth, [1, 2, 3, 4, 5], k))
        np.testing.assert_array_equal(native.arg_topk(s, k), ref.arg_topk(s, k))


def test_score_chain_close_to_blas():
    rng = np.random.RandomState(3)
    P = (rng.randn(50, 64) * 0.1).astype(np.float32)
    Q = (rng.randn(300, 64) * 0.1).astype(np.float32)
    users = rng.randint(0, 50, 20).astype(np.int32)
    S = native.score_gemm(P, users, Q)
    ref64 = P[users].astype(np.float64) @ Q.astype(np.float64).T
    assert np.abs(S - ref64).max() < 5e-7
    assert np.abs(S - P[users] @ Q.T).max() < 5e-7     # any fp32 order agrees to a few ulp


# ---------------------------------------------------------------- TF-half restatement
def _toy_graph(rng, U=30, I=40, per_user=5):
    users = np.repeat(np.arange(U), per_user)
    items = np.concatenate([rng.choice(I, per_user, replace=False) for _ in range(U)])
    return users, items, U, I


def test_lightgcn_adjacency_matches_hand_computation():
    rng = np.random.RandomState(0)
    users, items, U, I = _toy_graph(rng)
    A = train.lightgcn_adjacency(users, items, U, I, "pre")
    dense = np.zeros((U + I, U + I), np.float32)
    dense[users, items + U] = 1
    dense[items + U, users] = 1
    deg = dense.sum(1)
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, deg ** -0.5, 0).astype(np.float32)
    expect = (dinv[:, None] * dense) * dinv[None, :]
    np.testing.assert_array_equal(A.toarray(), expect)           # two fp32 multiplies, as the reference
    assert (A != A.T).nnz == 0                                     # symmetric ('pre')
    An = train.lightgcn_adjacency(users, items, U, I, "norm")
    np.testing.assert_allclose(np.asarray(An.sum(1)).ravel(), 1.0, rtol=1e-6)


def test_mf_and_lightgcn_fp32_track_fp64_twin():
    rng = np.random.RandomState(1)
    users, items, U, I = _toy_graph(rng)
    d, B = 16, 64
    A32 = train.lightgcn_adjacency(users, items, U, I, "pre")
    E32 = (rng.randn(U + I, d) * 0.1).astype(np.float32)
    E64 = E32.astype(np.float64)
    m32, v32 = np.zeros_like(E32), np.zeros_like(E32)
    m64, v64 = np.zeros_like(E64), np.zeros_like(E64)
    a32, a64 = train.Adam(0.01), train.Adam(0.01, dtype=np.float64)
    for step in range(5):
        bu = rng.randint(0, U, B); bp = rng.randint(0, I, B); bn = rng.randint(0, I, B)
        l32 = train.lightgcn_step(A32, A32.T.tocsr(), E32, m32, v32, U, 3, bu, bp, bn, 1e-3, a32)
        l64 = train.lightgcn_step(A32.astype(np.float64), A32.T.tocsr().astype(np.float64), E64,
                                  m64, v64, U, 3, bu, bp, bn, 1e-3, a64)
        assert abs(l32[0] - l64[0]) < 1e-5 * max(1, abs(l64[0]))
    assert np.abs(E32 - E64).max() < 1e-5
    # numerical gradient check of the restated backward pass (fp64)
    bu = rng.randint(0, U, 8); bp = rng.randint(0, I, 8); bn = rng.randint(0, I, 8)
    A64 = A32.astype(np.float64)
    mf, emb, g = train.lightgcn_loss_and_grad(A64, A64.T.tocsr(), E64, U, 2, bu, bp, bn, 1e-2)
    for (r, c) in [(int(bu[0]), 3), (U + int(bp[1]), 5), (7, 0)]:
        Ep = E64.copy(); Ep[r, c] += 1e-6
        Em = E64.copy(); Em[r, c] -= 1e-6
        fp = sum(train.lightgcn_loss_and_grad(A64, A64.T.tocsr(), Ep, U, 2, bu, bp, bn, 1e-2)[:2])
        fm = sum(train.lightgcn_loss_and_grad(A64, A64.T.tocsr(), Em, U, 2, bu, bp, bn, 1e-2)[:2])
        assert abs((fp - fm) / 2e-6 - g[r, c]) < 1e-6
    # MF: gradient check + duplicate indices are summed
    P = rng.randn(U, d) * 0.1; Q = rng.randn(I, d) * 0.1
    bu = np.array([0, 0, 1, 2]); bp = np.array([3, 3, 4, 5]); bn = np.array([6, 7, 3, 3])
    loss, dP, dQ = train.mf_loss_and_grads(P, Q, bu, bp, bn, 0.05)
    Pp = P.copy(); Pp[0, 2] += 1e-6
    lp = train.mf_loss_and_grads(Pp, Q, bu, bp, bn, 0.05)[0]
    assert abs((lp - loss) / 1e-6 - dP[0, 2]) < 1e-5
    Qp = Q.copy(); Qp[3, 1] += 1e-6
    lq = train.mf_loss_and_grads(P, Qp, bu, bp, bn, 0.05)[0]
    assert abs((lq - loss) / 1e-6 - dQ[3, 1]) < 1e-5


def test_tf_sparse_adam_sweeps_untouched_rows():
    """TF-1.12 semantics: rows without gradient still decay and move."""
    P = np.ones((4, 2), np.float32); m = np.full((4, 2), 0.5, np.float32); v = np.full((4, 2), 0.25, np.float32)
    g = np.zeros_like(P); g[1] = 1.0
    ad = train.Adam(0.1)
    train.Adam.sparse_swept(ad, P, m, v, g)
    assert np.all(m[0] == np.float32(0.5) * np.float32(0.9)) and np.all(P[0] < 1.0)
    assert np.all(m[1] > m[0])


def test_torch_cpu_step_port_tracks_the_numpy_oracle():
    """oracle.train_torch (bench.py's nproc-thread CPU baseline leg) is the same step as
    oracle.train.lightgcn_step: four steps, losses within 1e-6 relative, tables within 1e-4 (fp32
    sums in a different association)."""
    from oracle import train as O
    from oracle.train_torch import TorchLightGCN
    rng = np.random.RandomState(0)
    U, I, d, L = 300, 200, 32, 3
    users = np.repeat(np.arange(U), 5)
    items = np.concatenate([rng.choice(I, 5, replace=False) for _ in range(U)])
    A = O.lightgcn_adjacency(users, items, U, I, "pre")
    E0 = rng.uniform(-0.1, 0.1, (U + I, d)).astype(np.float32)
    t = TorchLightGCN(A, E0, U, L, 0.01, 1e-3, 2)
    e, m, v, ad = E0.copy(), np.zeros_like(E0), np.zeros_like(E0), O.Adam(0.01)
    for _ in range(4):
        u, p, n = rng.randint(0, U, 256), rng.randint(0, I, 256), rng.randint(0, I, 256)
        a, b = t.step(u, p, n), O.lightgcn_step(A, A, e, m, v, U, L, u, p, n, 1e-3, ad)
        assert abs(a[0] - b[0]) <= 1e-6 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-5 * abs(b[1])
    assert np.abs(t.E.numpy() - e).max() < 1e-4


@pytest.mark.skipif(not __import__("oracle.ref", fromlist=["x"]).sampler_module(),
                    reason="oracle/_ref (reference sampler modules) not built")
def test_reference_sampler_module_runs_an_epoch():
    """oracle.ref.sampler_module() = the reference's data/sampler.py compiled unchanged: one epoch
    has the structure SURVEY Appendix A records (aligned triplets, negatives outside the user's
    train items, last short batch kept)."""
    from oracle import ref
    mod = ref.sampler_module()
    d = {0: [1, 3], 2: [0], 5: [2, 4, 6, 7]}

    class DS:
        num_items = 10

        @staticmethod
        def get_user_train_dict():
            return d
    s = mod.PairwiseSampler(DS, neg_num=1, batch_size=4, shuffle=False)
    batches = list(s)
    assert len(s) == 2 and [len(b[0]) for b in batches] == [4, 3]
    users = sum((b[0] for b in batches), [])
    pos = sum((b[1] for b in batches), [])
    neg = sum((b[2] for b in batches), [])
    assert users == [0, 0, 2, 5, 5, 5, 5] and pos == [1, 3, 0, 2, 4, 6, 7]
    assert all(n not in d[u] and 0 <= n < 10 for u, n in zip(users, neg))
