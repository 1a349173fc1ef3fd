"""HIP training path against oracle.train: CSR SpMM (+ fused epilogues), BPR heads, TF-Adam
sweeps, and whole MF / LightGCN steps.

Tolerances (fp32, stated per north_star): loss and embeddings within 1e-5; kernels whose
operation order is identical to the oracle's (SpMM single-segment rows, Adam) bit-exact."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from neurec_amd import engine
    return engine


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _graph(rng, U, I, per_user_lo, per_user_hi, hubs=0):
    users, items = [], []
    for u in range(U):
        n = rng.randint(per_user_lo, per_user_hi)
        it = rng.choice(I, n, replace=False)
        users += [u] * n; items += it.tolist()
    for h in range(hubs):                              # hub items interacted by most users
        uu = rng.choice(U, int(U * 0.9), replace=False)
        users += uu.tolist(); items += [h] * len(uu)
    m = sp.csr_matrix((np.ones(len(users), np.float32), (users, items)), shape=(U, I))
    m.data[:] = 1.0
    coo = m.tocoo()
    return coo.row, coo.col


@pytest.mark.parametrize("d", [16, 32, 64, 128])
def test_spmm_matches_scipy_rowwise(eng, d):
    import torch
    from oracle import train
    rng = np.random.RandomState(d)
    U, I = 700, 500
    ur, ic = _graph(rng, U, I, 0, 30, hubs=2)         # 2 hub columns: rows of ~630 nnz -> split rows
    A = train.lightgcn_adjacency(ur, ic, U, I, "pre")
    X = rng.randn(U + I, d).astype(np.float32)
    csr = eng.SpmmCSR.from_scipy(A)
    assert csr.n_split_rows >= 2
    Y = torch.empty_like(_dev(X))
    csr.matmul(_dev(X), out=Y)
    got, want = Y.cpu().numpy(), train.spmm_rowwise(A, X)
    short = np.diff(A.indptr) <= csr.exact_row_nnz(d)               # 64 (d=64 lane-group kernel) / 256
    np.testing.assert_array_equal(got[short], want[short])          # same order, same roundings
    ref64 = A.astype(np.float64) @ X.astype(np.float64)
    assert np.abs(got - ref64).max() < 2e-5                          # split rows: reassociated, not wrong
    assert np.abs(got[~short] - want[~short]).max() < 1e-5
    # fused epilogues: addend and running sum (in-place sum_in == sum_out allowed)
    add, acc = rng.randn(U + I, d).astype(np.float32), rng.randn(U + I, d).astype(np.float32)
    Y2, S2 = torch.empty_like(Y), _dev(acc)
    csr.matmul(_dev(X), out=Y2, addend=_dev(add), sum_in=S2, sum_out=S2)
    np.testing.assert_array_equal(Y2.cpu().numpy()[short], (want + add)[short])
    np.testing.assert_array_equal(S2.cpu().numpy()[short], (acc + (want + add))[short])
    S3 = torch.empty_like(Y)
    csr.matmul(_dev(X), out=None, sum_in=_dev(acc), sum_out=S3)     # sum only (last forward layer)
    np.testing.assert_array_equal(S3.cpu().numpy()[short], (acc + want)[short])
    # non-symmetric ('norm') adjacency and its transpose
    An = train.lightgcn_adjacency(ur, ic, U, I, "norm")
    Ant = An.T.tocsr(); Ant.sort_indices()
    for M in (An, Ant):
        c = eng.SpmmCSR.from_scipy(M)
        Yn = torch.empty_like(Y); c.matmul(_dev(X), out=Yn)
        sh = np.diff(M.indptr) <= c.exact_row_nnz(d)
        np.testing.assert_array_equal(Yn.cpu().numpy()[sh], train.spmm_rowwise(M, X)[sh])
        assert np.abs(Yn.cpu().numpy() - train.spmm_rowwise(M, X)).max() < 1e-5


@pytest.mark.parametrize("d", [16, 64, 128])
def test_spmm_sums_a_row_in_its_storage_order(eng, d):
    """`gcmc` as the reference hands it to TF: every row's columns DESCENDING (graph.lightgcn_adjacency with
    tf_order=True).  The kernels walk a row's entries as stored, so the sums are those of the descending walk —
    bit for bit on rows that are not cut into segments — and differ in the last ulp from the ascending walk."""
    import torch
    from neurec_amd.graph import lightgcn_adjacency
    rng = np.random.RandomState(d + 1)
    U, I = 700, 500
    ur, ic = _graph(rng, U, I, 0, 30, hubs=1)
    A_desc = lightgcn_adjacency(ur, ic, U, I, "gcmc", tf_order=True)
    A_asc = lightgcn_adjacency(ur, ic, U, I, "gcmc")
    assert np.all(np.diff(A_desc.indices)[np.diff(np.repeat(np.arange(U + I), np.diff(A_desc.indptr))) == 0] < 0)
    X = rng.randn(U + I, d).astype(np.float32)
    csr = eng.SpmmCSR.from_scipy(A_desc, keep_order=True)
    Y = torch.empty_like(_dev(X))
    csr.matmul(_dev(X), out=Y)
    got = Y.cpu().numpy()

    def walk(A):                                   # sequential fp32 accumulation in storage order
        out = np.zeros((A.shape[0], d), np.float32)
        for r in range(A.shape[0]):
            acc = np.zeros(d, np.float32)
            for p in range(A.indptr[r], A.indptr[r + 1]):
                acc = acc + A.data[p] * X[A.indices[p]]
            out[r] = acc
        return out
    want_desc, want_asc = walk(A_desc), walk(A_asc)
    short = np.diff(A_desc.indptr) <= csr.exact_row_nnz(d)
    np.testing.assert_array_equal(got[short], want_desc[short])
    assert (want_desc[short] != want_asc[short]).any()          # the order is observable
    assert np.abs(got - want_asc).max() < 1e-5


@pytest.mark.parametrize("d", [128, 256])
def test_spmm_lane_group_kernel_wide_rows(eng, d):
    """the persistent lane-group kernel at d = 128 / 256 (2 / 1 rows per load instruction), forced:
    same contract as the work-item kernel, rows of <= 64 nnz in strict order"""
    import torch
    from oracle import train
    rng = np.random.RandomState(d + 5)
    U, I = 600, 450
    ur, ic = _graph(rng, U, I, 0, 30, hubs=2)
    A = train.lightgcn_adjacency(ur, ic, U, I, "pre")
    X = rng.randn(U + I, d).astype(np.float32)
    add, acc = rng.randn(U + I, d).astype(np.float32), rng.randn(U + I, d).astype(np.float32)
    csr = eng.SpmmCSR.from_scipy(A, split_row=U)
    assert csr.ensure_schedule(d, force=True) and csr.exact_row_nnz(d) == 64
    Y, S = torch.empty_like(_dev(X)), torch.empty_like(_dev(X))
    csr.matmul(_dev(X), out=Y, addend=_dev(add), sum_in=_dev(acc), sum_out=S)
    want = train.spmm_rowwise(A, X)
    short = np.diff(A.indptr) <= 64
    np.testing.assert_array_equal(Y.cpu().numpy()[short], (want + add)[short])
    np.testing.assert_array_equal(S.cpu().numpy()[short], (acc + (want + add))[short])
    assert np.abs(Y.cpu().numpy() - (want + add)).max() < 1e-5
    flag = np.zeros(U + I, np.uint8); flag[rng.choice(U + I, 120, replace=False)] = 1
    Xz = X * flag[:, None]
    y_full, y_mask = torch.empty_like(Y), torch.empty_like(Y)
    csr.matmul(_dev(Xz), out=y_full)
    csr.matmul(_dev(Xz), out=y_mask, x_row_nonzero=_dev(flag))
    np.testing.assert_array_equal(y_mask.cpu().numpy(), y_full.cpu().numpy())
    part = torch.full((U + I, d), 9.0, device="cuda")
    csr.matmul(_dev(X), out=part, y_row_wanted=_dev(flag))
    got = part.cpu().numpy()
    full = torch.empty_like(Y); csr.matmul(_dev(X), out=full)
    np.testing.assert_array_equal(got[flag == 1], full.cpu().numpy()[flag == 1])
    assert np.all(got[flag == 0] == 9.0)


def test_spmm_empty_rows_and_tiny(eng):
    import torch
    A = sp.csr_matrix(([0.5, 2.0, -1.0], ([0, 0, 3], [1, 3, 0])), shape=(5, 5), dtype=np.float32)
    X = np.arange(5 * 64, dtype=np.float32).reshape(5, 64) / 7
    c = eng.SpmmCSR.from_scipy(A)
    Y = torch.full((5, 64), 7.0, device="cuda")
    c.matmul(_dev(X), out=Y)
    np.testing.assert_array_equal(Y.cpu().numpy(), (A @ X).astype(np.float32))   # empty rows -> 0


def test_adam_sweeps_bit_exact(eng):
    from oracle import train
    rng = np.random.RandomState(0)
    n = 70839 * 3 + 1                                   # not a multiple of 4: exercises the tail
    for sparse in (True, False):
        var = rng.randn(n).astype(np.float32); m = (rng.randn(n) * 1e-2).astype(np.float32)
        v = (rng.rand(n) * 1e-3).astype(np.float32); g = (rng.randn(n) * 0.1).astype(np.float32)
        if sparse:
            g[rng.rand(n) < 0.9] = 0
        dv, dm, dvv, dg = _dev(var), _dev(m), _dev(v), _dev(g)
        st, ad = eng.AdamState(0.001), train.Adam(0.001)
        for _ in range(4):
            st.advance(); ad.advance()
        assert st.alpha() == ad.alpha()
        if sparse:
            eng.adam_sparse(dv, dm, dvv, dg, st); ad.sparse_swept(var, m, v, g)
        else:
            eng.adam_dense(dv, dm, dvv, dg, st, clear_grad=True); ad.dense(var, m, v, g)
        np.testing.assert_array_equal(dm.cpu().numpy(), m)
        np.testing.assert_array_equal(dvv.cpu().numpy(), v)
        np.testing.assert_array_equal(dv.cpu().numpy(), var)
        assert not dg.cpu().numpy().any()               # gradient buffer cleared for the next step


@pytest.mark.parametrize("d", [16, 64, 128])
def test_bpr_mf_step_tracks_oracle(eng, d):
    import torch
    from neurec_amd.trainer import MFEngine
    from oracle import train
    rng = np.random.RandomState(d)
    U, I, B, reg, lr = 300, 400, 512, 0.01, 0.001
    P = (rng.randn(U, d) * 0.01).astype(np.float32); Q = (rng.randn(I, d) * 0.01).astype(np.float32)
    mf = MFEngine(P, Q, lr, reg, B)
    oP, oQ = P.copy(), Q.copy()
    om = [np.zeros_like(P), np.zeros_like(P), np.zeros_like(Q), np.zeros_like(Q)]
    P64, Q64 = P.astype(np.float64), Q.astype(np.float64)
    om64 = [np.zeros_like(P64), np.zeros_like(P64), np.zeros_like(Q64), np.zeros_like(Q64)]
    ad, ad64 = train.Adam(lr), train.Adam(lr, dtype=np.float64)
    loss2 = torch.zeros(2, device="cuda")
    for step in range(6):
        bu = rng.randint(0, U, B).astype(np.int32); bp = rng.randint(0, I, B).astype(np.int32)
        bn = rng.randint(0, I, B).astype(np.int32)
        if step == 0:
            bu[:50] = bu[0]; bp[:20] = bp[0]; bn[20:40] = bp[0]       # heavy duplicates in a batch
        mf.step(_dev(bu), _dev(bp), _dev(bn), loss2)
        want = train.mf_step(oP, oQ, om[0], om[1], om[2], om[3], bu, bp, bn, reg, ad)
        want64 = train.mf_step(P64, Q64, om64[0], om64[1], om64[2], om64[3], bu, bp, bn, reg, ad64)
        got = float(loss2.sum().item())
        assert abs(got - want64) <= 1e-5 * abs(want64), (step, got, want, want64)
        assert abs(got - want) <= 1e-5 * abs(want)
    assert np.abs(mf.P.cpu().numpy() - P64).max() < 1e-5 and np.abs(mf.Q.cpu().numpy() - Q64).max() < 1e-5
    assert np.abs(mf.P.cpu().numpy() - oP).max() < 2e-6


@pytest.mark.parametrize("adj_type,L,d", [("pre", 3, 64), ("pre", 1, 16), ("norm", 2, 64), ("pre", 0, 64),
                                          # every built width through the native step with L >= 2 (r03: the
                                          # fused-Adam last hop was taken for any width that had a lane-group
                                          # schedule, but exists at d = 64 only: d = 16 / 32 raised)
                                          ("pre", 3, 16), ("pre", 2, 32), ("pre", 3, 128), ("norm", 3, 16),
                                          # widths that are not built run zero-padded to the next one that is
                                          ("pre", 3, 50), ("pre", 2, 20), ("norm", 2, 100)])
def test_lightgcn_step_tracks_oracle(eng, adj_type, L, d):
    import torch
    from neurec_amd.trainer import LightGCNEngine
    from oracle import train
    rng = np.random.RandomState(L * 10 + d)
    U, I, B, reg, lr = 400, 300, 256, 1e-3, 0.01
    ur, ic = _graph(rng, U, I, 1, 25, hubs=1)
    A = train.lightgcn_adjacency(ur, ic, U, I, adj_type)
    At = A.T.tocsr(); At.sort_indices()
    lim = np.sqrt(6.0 / (U + d))
    E0 = rng.uniform(-lim, lim, (U + I, d)).astype(np.float32)
    lg = LightGCNEngine(A, U, I, E0, L, lr, reg, B, adj_t_csr=None if adj_type == "pre" else At)
    o32, m32, v32 = E0.copy(), np.zeros_like(E0), np.zeros_like(E0)
    o64 = E0.astype(np.float64); m64, v64 = np.zeros_like(o64), np.zeros_like(o64)
    A64, At64 = A.astype(np.float64), At.astype(np.float64)
    ad, ad64 = train.Adam(lr), train.Adam(lr, dtype=np.float64)
    # forward only first: E* against the oracle's propagation
    eu, ei = lg.final_embeddings()
    want_star, _ = train.lightgcn_propagate(A, E0, L)
    got_star = np.concatenate([eu.cpu().numpy(), ei.cpu().numpy()])
    assert np.abs(got_star - want_star).max() < 1e-6
    loss2 = torch.zeros(2, device="cuda")
    for step in range(5):
        bu = rng.randint(0, U, B).astype(np.int32); bp = rng.randint(0, I, B).astype(np.int32)
        bn = rng.randint(0, I, B).astype(np.int32)
        lg.step(_dev(bu), _dev(bp), _dev(bn), loss2)
        w32 = train.lightgcn_step(A, At, o32, m32, v32, U, L, bu, bp, bn, reg, ad)
        w64 = train.lightgcn_step(A64, At64, o64, m64, v64, U, L, bu, bp, bn, reg, ad64)
        got = loss2.cpu().numpy()
        assert abs(got[0] - w64[0]) <= 1e-5 * abs(w64[0]), (step, got, w32, w64)
        assert abs(got[1] - w64[1]) <= 1e-5 * max(abs(w64[1]), 1e-3)
    gotE = lg.E0.cpu().numpy()
    assert lg.d_real == d and not gotE[:, d:].any()          # padded columns (widths that are not built) stay zero
    gotE = gotE[:, :d]
    assert np.abs(gotE - o64).max() < 1e-5, np.abs(gotE - o64).max()
    assert np.abs(gotE - o32).max() < 1e-5
    assert not lg.Greg.cpu().numpy().any() and not lg.Gstar.cpu().numpy().any()   # buffers re-armed


def test_lightgcn_training_improves_ndcg_end_to_end(eng):
    """Sampler -> LightGCN steps -> full-rank evaluator, all on the device: learning happens."""
    import torch
    from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, LightGCNEngine
    from oracle import train
    rng = np.random.RandomState(5)
    U, I, d, L = 300, 200, 64, 2
    # planted structure: user u likes items of its cluster
    ur, ic = [], []
    for u in range(U):
        base = (u % 10) * 20
        its = base + rng.choice(20, 9, replace=False)
        ur += [u] * 9; ic += its.tolist()
    ur, ic = np.array(ur), np.array(ic)
    is_test = np.arange(len(ur)) % 9 == 0
    tr = sp.csr_matrix((np.ones((~is_test).sum()), (ur[~is_test], ic[~is_test])), shape=(U, I))
    te = sp.csr_matrix((np.ones(is_test.sum()), (ur[is_test], ic[is_test])), shape=(U, I))
    trc, tec = eng.DeviceCSR.from_scipy(tr), eng.DeviceCSR.from_scipy(te)
    A = train.lightgcn_adjacency(ur[~is_test], ic[~is_test], U, I, "pre")
    lim = np.sqrt(6.0 / (U + d))
    lg = LightGCNEngine(A, U, I, rng.uniform(-lim, lim, (U + I, d)).astype(np.float32), L, 0.01,
                        1e-3, 512)
    sampler = BprEpochSampler(trc, I, batch_size=512, seed=2018)
    ev = FullRankEvaluator(trc, tec, [4], 10, batch_rows=512)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    loss2 = torch.zeros(2, device="cuda")

    def ndcg10():
        eu, ei = lg.final_embeddings()
        return ev.evaluate_factors(eu.contiguous(), ei.contiguous(), users)[9]
    before = ndcg10()
    for _ in range(30):
        for bu, bp, bn in sampler.batches():
            lg.step(bu, bp, bn, loss2)
    after = ndcg10()
    assert after > before + 0.2, (before, after)


@pytest.mark.parametrize("d", [64, 128])
def test_spmm_row_subset_and_masked_variants_equal_full_product(eng, d):
    """The two work-skipping variants used inside a training step return exactly the rows /
    values of the full product (they only skip work whose result is unused or zero)."""
    import torch
    from oracle import train
    rng = np.random.RandomState(d + 1)
    U, I = 900, 700
    ur, ic = _graph(rng, U, I, 0, 40, hubs=3)          # hub item rows ~800 nnz: 4 segments
    A = train.lightgcn_adjacency(ur, ic, U, I, "pre")
    N = U + I
    X = rng.randn(N, d).astype(np.float32)
    csr = eng.SpmmCSR.from_scipy(A)
    acc = rng.randn(N, d).astype(np.float32)
    full = torch.empty(N, d, device="cuda")
    csr.matmul(_dev(X), out=None, sum_in=_dev(acc), sum_out=full)
    rows = np.concatenate([rng.randint(0, N, 500), [U, U + 1, U + 2, 0, 0, U]]).astype(np.int32)
    part = torch.full((N, d), 123.0, device="cuda")
    csr.matmul_rows(_dev(X), _dev(rows), sum_in=_dev(acc), sum_out=part)
    got, want = part.cpu().numpy(), full.cpu().numpy()
    lens = np.diff(A.indptr)
    same_order = rows[lens[rows] <= csr.exact_row_nnz(d)]            # longer rows: segments re-associated
    np.testing.assert_array_equal(got[same_order], want[same_order])
    assert np.abs(got[rows] - want[rows]).max() < 1e-5
    untouched = np.setdiff1d(np.arange(N), rows)
    assert np.all(got[untouched] == 123.0)
    # masked: only flagged rows of X are non-zero
    flag = np.zeros(N, np.uint8); flag[rng.choice(N, 150, replace=False)] = 1; flag[U] = 1
    Xz = X * flag[:, None]
    add = rng.randn(N, d).astype(np.float32)
    y_full, y_mask = torch.empty(N, d, device="cuda"), torch.empty(N, d, device="cuda")
    csr.matmul(_dev(Xz), out=y_full, addend=_dev(add))
    csr.matmul(_dev(Xz), out=y_mask, addend=_dev(add), x_row_nonzero=_dev(flag))
    np.testing.assert_array_equal(y_mask.cpu().numpy(), y_full.cpu().numpy())
    # wanted-rows filter (hub rows U, U+1 included): produced rows identical, others untouched
    wanted = np.zeros(N, np.uint8); wanted[rows] = 1
    part2 = torch.full((N, d), 321.0, device="cuda")
    csr.matmul(_dev(X), out=None, sum_in=_dev(acc), sum_out=part2, y_row_wanted=_dev(wanted))
    got2 = part2.cpu().numpy()
    np.testing.assert_array_equal(got2[wanted == 1], want[wanted == 1])
    assert np.all(got2[wanted == 0] == 321.0)
    # both filters at once
    y_both = torch.full((N, d), 5.0, device="cuda")
    csr.matmul(_dev(Xz), out=y_both, addend=_dev(add), x_row_nonzero=_dev(flag),
               y_row_wanted=_dev(wanted))
    gb = y_both.cpu().numpy()
    np.testing.assert_array_equal(gb[wanted == 1], y_full.cpu().numpy()[wanted == 1])
    assert np.all(gb[wanted == 0] == 5.0)


def test_native_step_equals_python_launch_sequence(eng):
    """csrc/step.hip only orders launches: it must leave exactly the same state as the
    spelled-out Python sequence (row gradients are summed in batch order: nothing is unordered)."""
    import torch
    from neurec_amd.trainer import LightGCNEngine, MFEngine
    from oracle import train
    rng = np.random.RandomState(77)
    U, I, d, L, B = 500, 400, 64, 3, 512
    ur, ic = _graph(rng, U, I, 1, 30, hubs=2)
    A = train.lightgcn_adjacency(ur, ic, U, I, "pre")
    E0 = rng.uniform(-0.1, 0.1, (U + I, d)).astype(np.float32)
    a, b = LightGCNEngine(A, U, I, E0, L, 0.01, 1e-3, B), LightGCNEngine(A, U, I, E0, L, 0.01, 1e-3, B)
    la, lb = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    for step in range(4):
        bu, bp, bn = (_dev(rng.randint(0, n, B).astype(np.int32)) for n in (U, I, I))
        a.step(bu, bp, bn, la)
        b.step_reference(bu, bp, bn, lb)
        assert float(la[0]) == float(lb[0]) and float(la[1]) == float(lb[1])
    np.testing.assert_array_equal(a.E0.cpu().numpy(), b.E0.cpu().numpy())
    assert a.adam.t == b.adam.t == 4
    # multi-GPU form of the step with an identity "all-reduce" == the single call
    c = LightGCNEngine(A, U, I, E0, L, 0.01, 1e-3, B)
    rng2 = np.random.RandomState(78)
    d2 = LightGCNEngine(A, U, I, E0, L, 0.01, 1e-3, B)
    for step in range(3):
        bu, bp, bn = (_dev(rng2.randint(0, n, B).astype(np.int32)) for n in (U, I, I))
        c.step(bu, bp, bn, None, grad_sync=lambda t: t)
        d2.step(bu, bp, bn, None)
    # the cut form adds G_0 + reg rows before Adam, the fused form inside it: same fp32 addition
    np.testing.assert_array_equal(c.E0.cpu().numpy(), d2.E0.cpu().numpy())
    P = (rng.randn(U, 32) * 0.01).astype(np.float32); Q = (rng.randn(I, 32) * 0.01).astype(np.float32)
    m1, m2 = MFEngine(P, Q, 0.001, 0.01, B), MFEngine(P, Q, 0.001, 0.01, B, lazy=False)   # lazy replay vs sweep
    for step in range(3):
        bu, bp, bn = (_dev(rng.randint(0, n, B).astype(np.int32)) for n in (U, I, I))
        m1.step(bu, bp, bn, la)
        m2.step_reference(bu, bp, bn, lb)
    np.testing.assert_array_equal(m1.P.cpu().numpy(), m2.P.cpu().numpy())
    np.testing.assert_array_equal(m1.Q.cpu().numpy(), m2.Q.cpu().numpy())
    assert float(la[0]) == float(lb[0]) and float(la[1]) == float(lb[1])


def test_spmm_with_fused_adam_epilogue_equals_two_passes(eng):
    """nrhip_spmm_csr_adam (last backward hop + ApplyAdam in one pass) == nrhip_spmm_csr followed by
    nrhip_adam_dense_tf2, bit for bit (same products, same sums, same Adam arithmetic)."""
    import ctypes as C
    import torch
    from neurec_amd._lib import call
    from oracle import train
    rng = np.random.RandomState(31)
    U, I, d = 900, 700, 64
    ur, ic = _graph(rng, U, I, 0, 30, hubs=2)
    A = train.lightgcn_adjacency(ur, ic, U, I, "pre")
    N = U + I
    csr = eng.SpmmCSR.from_scipy(A, split_row=U)
    assert csr.ensure_schedule(d)
    X, H, Gb = (_dev(rng.randn(N, d).astype(np.float32) * s) for s in (1.0, 0.1, 0.01))
    var0, m0, v0 = rng.randn(N, d).astype(np.float32), rng.randn(N, d).astype(np.float32) * 0.01, \
        (rng.rand(N, d).astype(np.float32) * 1e-3)
    st = eng.AdamState(0.01); st.advance(); st.advance()
    # two passes
    var_a, m_a, v_a = _dev(var0), _dev(m0), _dev(v0)
    Y = torch.empty(N, d, device="cuda")
    csr.matmul(X, out=Y, addend=H)
    eng.adam_dense2(var_a, m_a, v_a, Y, Gb, st)
    # one pass
    var_b, m_b, v_b = _dev(var0), _dev(m0), _dev(v0)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    call("nrhip_spmm_csr_adam", csr.plan, ptr(csr.indices), ptr(csr.vals), ptr(X), d, ptr(H), ptr(Gb),
         ptr(var_b), ptr(m_b), ptr(v_b), float(st.alpha()), float(st.beta1), float(st.beta2),
         float(st.eps), 0, C.c_void_p(0), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    for a, b in ((var_a, var_b), (m_a, m_b), (v_a, v_b)):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    # clear_consumed: same update, and the sparse inputs / flags come back zeroed
    var_c, m_c, v_c = _dev(var0), _dev(m0), _dev(v0)
    flags = np.zeros(N, np.uint8); flags[rng.choice(N, 200, replace=False)] = 1
    Hs = H * _dev(flags[:, None].astype(np.float32)); Gs = Gb * _dev(flags[:, None].astype(np.float32))
    var_d, m_d, v_d = _dev(var0), _dev(m0), _dev(v0)
    Y2 = torch.empty(N, d, device="cuda"); csr.matmul(X, out=Y2, addend=Hs)
    eng.adam_dense2(var_d, m_d, v_d, Y2, Gs, st)
    fl = _dev(flags)
    call("nrhip_spmm_csr_adam", csr.plan, ptr(csr.indices), ptr(csr.vals), ptr(X), d, ptr(Hs), ptr(Gs),
         ptr(var_c), ptr(m_c), ptr(v_c), float(st.alpha()), float(st.beta1), float(st.beta2),
         float(st.eps), 1, ptr(fl), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    np.testing.assert_array_equal(var_c.cpu().numpy(), var_d.cpu().numpy())
    assert not Hs.cpu().numpy().any() and not Gs.cpu().numpy().any() and not fl.cpu().numpy().any()
    # without the schedule the entry point refuses (the step driver then runs the two passes)
    plain = eng.SpmmCSR.from_scipy(A)
    with pytest.raises(NotImplementedError):
        call("nrhip_spmm_csr_adam", plain.plan, ptr(plain.indices), ptr(plain.vals), ptr(X), d, ptr(H),
             ptr(Gb), ptr(var_b), ptr(m_b), ptr(v_b), 0.01, 0.9, 0.999, 1e-8, 0, C.c_void_p(0), C.c_void_p(0))


@pytest.mark.gpu
def test_spmm_column_masked_kernel_variants(eng, monkeypatch):
    """spmm_colmasked_kernel (LDS-staged slice, sub-lists compacted to the surviving columns) against
    the oracle's row-wise product and against the general masked kernel: empty rows, hub rows split
    into segments, an addend that is the operand itself (its zero rows are not read), running sum."""
    import torch
    from oracle import train
    rng = np.random.RandomState(77)
    U, I, d = 1100, 800, 64
    ur, ic = _graph(rng, U, I, 0, 50, hubs=3)
    keep = ~np.isin(ur, [5, 6, 700, U - 1]) & ~np.isin(ic, [9, 10, I - 1])     # users / items nobody touches
    ur, ic = ur[keep], ic[keep]
    A = train.lightgcn_adjacency(ur, ic, U, I, "pre")
    N = U + I
    assert (np.diff(A.indptr) == 0).sum() >= 7 and (np.diff(A.indptr) > 128).any()
    flag = np.zeros(N, np.uint8); flag[rng.choice(N, 300, replace=False)] = 1; flag[[U, U + 1]] = 1
    X = (rng.randn(N, d).astype(np.float32)) * flag[:, None]
    add, acc = rng.randn(N, d).astype(np.float32), rng.randn(N, d).astype(np.float32)
    want = train.spmm_rowwise(A, X)
    short = np.diff(A.indptr) <= 64
    outs = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("NEUREC_SPMM_MASKED_FAST", fast)
        csr = eng.SpmmCSR.from_scipy(A, split_row=U)
        assert csr.ensure_schedule(d)
        Xd, fl = _dev(X), _dev(flag)
        y0 = torch.full((N, d), 9.0, device="cuda")
        csr.matmul(Xd, out=y0, x_row_nonzero=fl)
        y1 = torch.full((N, d), 9.0, device="cuda")
        csr.matmul(Xd, out=y1, addend=Xd, x_row_nonzero=fl)              # H + A·H, the step's use
        y2, s2 = torch.full((N, d), 9.0, device="cuda"), torch.full((N, d), 9.0, device="cuda")
        csr.matmul(Xd, out=y2, addend=_dev(add), sum_in=_dev(acc), sum_out=s2, x_row_nonzero=fl)
        outs[fast] = [t.cpu().numpy() for t in (y0, y1, y2, s2)]
    y0, y1, y2, s2 = outs["1"]
    np.testing.assert_array_equal(y0[short], want[short])               # same order, same roundings
    assert np.abs(y0 - want).max() < 1e-5
    np.testing.assert_array_equal(y1[short], (want + X)[short])
    np.testing.assert_array_equal(y2[short], (want + add)[short])
    np.testing.assert_array_equal(s2, acc + y2)
    for a, b in zip(outs["1"], outs["0"]):                              # and bit-identical to the general kernel
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("nnz_cap", [0, 256])
def test_spmm_wanted_rows_kernel(eng, monkeypatch, nnz_cap):
    """spmm_wanted_rows_kernel (row-masked hop on the dealt schedule, sub-lists staged in LDS) against
    the oracle and the general masked kernel; nnz_cap=256 forces the chunked path (wanted sub-lists
    outgrow the staging buffer and are re-dealt)."""
    import torch
    from oracle import train
    rng = np.random.RandomState(78)
    U, I, d = 1100, 800, 64
    ur, ic = _graph(rng, U, I, 0, 50, hubs=3)
    keep = ~np.isin(ur, [5, 6, 700, U - 1]) & ~np.isin(ic, [9, 10, I - 1])
    ur, ic = ur[keep], ic[keep]
    A = train.lightgcn_adjacency(ur, ic, U, I, "pre")
    N = U + I
    lens = np.diff(A.indptr)
    X = rng.randn(N, d).astype(np.float32)
    add, acc = rng.randn(N, d).astype(np.float32), rng.randn(N, d).astype(np.float32)
    wanted = np.zeros(N, np.uint8)
    wanted[rng.choice(N, 400, replace=False)] = 1
    wanted[[U, U + 1, U + 2, 5, U + 9]] = 1                      # the hub rows and two empty rows
    if nnz_cap:                                                  # many sub-lists in few workgroups
        wanted[np.argsort(-lens)[:300]] = 1
    want = train.spmm_rowwise(A, X)
    short = (lens <= 64) & (wanted == 1)
    outs = {}
    for fast in ("1", "0"):
        monkeypatch.setenv("NEUREC_SPMM_MASKED_FAST", fast)
        if nnz_cap:
            monkeypatch.setenv("NEUREC_SPMM_WANTED_NNZ_CAP", str(nnz_cap))
        csr = eng.SpmmCSR.from_scipy(A, split_row=U)
        assert csr.ensure_schedule(d)
        Xd, wd = _dev(X), _dev(wanted)
        y0 = torch.full((N, d), 7.0, device="cuda")
        csr.matmul(Xd, out=y0, y_row_wanted=wd)
        s1 = torch.full((N, d), 7.0, device="cuda")
        csr.matmul(Xd, out=None, sum_in=_dev(acc), sum_out=s1, y_row_wanted=wd)     # the step's use
        y2, s2 = torch.full((N, d), 7.0, device="cuda"), torch.full((N, d), 7.0, device="cuda")
        csr.matmul(Xd, out=y2, addend=_dev(add), sum_in=_dev(acc), sum_out=s2, y_row_wanted=wd)
        outs[fast] = [t.cpu().numpy() for t in (y0, s1, y2, s2)]
    y0, s1, y2, s2 = outs["1"]
    np.testing.assert_array_equal(y0[short], want[short])
    assert np.abs(y0 - want)[wanted == 1].max() < 1e-5
    assert np.all(y0[wanted == 0] == 7.0) and np.all(s1[wanted == 0] == 7.0)
    np.testing.assert_array_equal(s1[wanted == 1], (acc + y0)[wanted == 1])
    np.testing.assert_array_equal(y2[short], (want + add)[short])
    np.testing.assert_array_equal(s2[wanted == 1], (acc + y2)[wanted == 1])
    for a, b in zip(outs["1"], outs["0"]):
        np.testing.assert_array_equal(a, b)
    # layer chain: ((sum_in + layer_a) + layer_b) + A·X on the wanted rows, terms optional
    import ctypes as C
    from neurec_amd._lib import call, lib
    monkeypatch.setenv("NEUREC_SPMM_MASKED_FAST", "1")
    csr = eng.SpmmCSR.from_scipy(A, split_row=U)
    assert csr.ensure_schedule(d) and lib.nrhip_spmm_plan_has_wanted(csr.plan, d) == 2
    la, lb = rng.randn(N, d).astype(np.float32), rng.randn(N, d).astype(np.float32)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    Xd, wd, accd, lad, lbd = _dev(X), _dev(wanted), _dev(acc), _dev(la), _dev(lb)
    w = wanted == 1
    for a_, b_, ref in ((None, None, acc + y0), (lad, None, (acc + la) + y0), (lad, lbd, ((acc + la) + lb) + y0)):
        out = torch.full((N, d), 7.0, device="cuda")
        call("nrhip_spmm_csr_wanted_layers", csr.plan, ptr(csr.indices), ptr(csr.vals), ptr(Xd), d, ptr(accd),
             ptr(a_), ptr(b_), ptr(out), ptr(wd), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        got = out.cpu().numpy()
        np.testing.assert_array_equal(got[w], ref[w])
        assert np.all(got[~w] == 7.0)
    # batch form: the triplets instead of flags; it publishes the flags and the row list itself
    B = 300
    bu, bp, bn = (rng.randint(0, U, B).astype(np.int32), rng.randint(0, I, B).astype(np.int32),
                  rng.randint(0, I, B).astype(np.int32))
    bp[:3] = [0, 1, 2]                                            # the hub items
    rows_ref = np.concatenate([bu, U + bp, U + bn])
    wb = np.zeros(N, bool); wb[rows_ref] = True
    yb = torch.full((N, d), 7.0, device="cuda")
    csr.matmul(Xd, out=yb, y_row_wanted=_dev(wb.astype(np.uint8)))
    ref = ((acc + la) + lb) + yb.cpu().numpy()
    out = torch.full((N, d), 7.0, device="cuda")
    flags = torch.zeros(N, dtype=torch.uint8, device="cuda")
    rows_out = torch.full((3 * B,), -1, dtype=torch.int32, device="cuda")
    bud, bpd, bnd = _dev(bu), _dev(bp), _dev(bn)                 # keep them alive across the call
    call("nrhip_spmm_csr_wanted_batch", csr.plan, ptr(csr.indices), ptr(csr.vals), ptr(Xd), d, ptr(accd),
         ptr(lad), ptr(lbd), ptr(out), ptr(bud), ptr(bpd), ptr(bnd), B, U, ptr(flags),
         ptr(rows_out), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[wb], ref[wb])
    assert np.all(got[~wb] == 7.0)
    np.testing.assert_array_equal(flags.cpu().numpy().astype(bool), wb)
    np.testing.assert_array_equal(rows_out.cpu().numpy(), rows_ref)
