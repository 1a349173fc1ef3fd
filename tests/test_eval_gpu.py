"""HIP evaluator (top-K select + exact tie path + metrics, train mask, fp32-MFMA scoring)
against the CPU oracle and the golden vectors of the compiled reference.
Integer/index results must be bit-exact; so must the fp32 metric values (same
float/double operation sequence) and the GEMM scores (same fmaf chain)."""
import numpy as np
import pytest

from conftest import golden_eval_cases, load_golden, truth_lists

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    from neurec_amd import engine
    assert torch.cuda.is_available()
    return engine


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _csr(eng, lists, n_cols):
    from oracle.native import lists_to_csr
    ptr, idx = lists_to_csr(lists)
    return eng.DeviceCSR(ptr, idx[:int(ptr[-1])], n_cols)


@pytest.mark.parametrize("case", golden_eval_cases())
@pytest.mark.parametrize("pad", [False, True])
def test_eval_scores_reproduces_reference_golden(eng, case, pad):
    import torch
    g = load_golden(case)
    scores, K = g["scores"], int(g["top_k"])
    rows, cols = scores.shape
    truth = _csr(eng, truth_lists(g["truth_ptr"], g["truth_idx"]), cols)
    if pad:      # 16-byte aligned rows -> float4 kernel; padding columns hold junk that must be ignored
        ld = (cols + 63) // 64 * 64
        buf = torch.full((rows, ld), 1e30, dtype=torch.float32, device="cuda")
        buf[:, :cols] = _dev(scores)
    else:        # odd leading dimension -> dword kernel
        buf = _dev(scores)
    out, topk, nex = eng.eval_scores(buf, truth, g["metrics"].tolist(), K, cols=cols,
                                     want_topk=True, want_exact_count=True)
    np.testing.assert_array_equal(out.cpu().numpy(), g["result"])
    if case in ("eval_ties", "eval_pop"):
        assert int(nex.item()) > 0          # the exact libstdc++-heap path was exercised
    if case == "eval_random":
        assert int(nex.item()) == 0         # tie-free rows never leave the parallel path
    at = eng.arg_topk(buf, K, cols=cols)
    np.testing.assert_array_equal(at.cpu().numpy(), g["arg_topk"])


@pytest.mark.parametrize("rows,cols,k", [(1, 20, 20), (3, 64, 1), (130, 257, 33), (64, 1025, 50),
                                         (17, 5000, 128), (9, 40981, 20)])
def test_eval_scores_matches_oracle_random_and_ties(eng, rows, cols, k):
    from oracle import native
    rng = np.random.RandomState(rows * 1000 + cols)
    for mode in ("gauss", "ties", "masked"):
        s = rng.randn(rows, cols).astype(np.float32)
        if mode == "ties":
            s = (np.round(s * 4) / 4).astype(np.float32)
        if mode == "masked":
            s[rng.rand(rows, cols) < 0.4] = -np.inf
        truth_l = [np.sort(rng.choice(cols, rng.randint(1, min(cols, 300) + 1), replace=False)).tolist()
                   for _ in range(rows)]
        want, want_topk = native.eval_matrix(s, truth_l, [1, 2, 3, 4, 5], k, want_topk=True)
        got, got_topk = eng.eval_scores(_dev(s), _csr(eng, truth_l, cols), [1, 2, 3, 4, 5], k,
                                        want_topk=True)
        np.testing.assert_array_equal(got_topk.cpu().numpy(), want_topk)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        if k <= cols:
            np.testing.assert_array_equal(eng.arg_topk(_dev(s), k).cpu().numpy(),
                                          native.arg_topk(s, k))


@pytest.mark.parametrize("rows,cols,k", [(40, 300, 20), (16, 5000, 10), (6, 40981, 20), (12, 1682, 50)])
def test_rows_with_nan_scores_are_ranked_as_the_reference_heap_ranks_them(eng, rows, cols, k):
    """A NaN score (a diverged model) is not a key of the parallel selection; in the reference it sits in the heap of
    std::partial_sort_copy like any value and every comparison with it is false (evaluate.h:38-42), so the answer is a
    function of the row's whole history.  Such rows take the exact path — the heap replayed, NaNs included — and come
    out as the reference evaluator's: whole NaN rows, NaNs sprinkled among scores, NaNs next to -inf (masked) entries."""
    from oracle import native
    rng = np.random.RandomState(rows + cols)
    for mode in ("rows", "sprinkled", "with_masked"):
        s = rng.randn(rows, cols).astype(np.float32)
        if mode == "rows":
            s[::3] = np.nan
        else:
            s[rng.rand(rows, cols) < 0.01] = np.nan
            s[0, :] = rng.randn(cols)                          # (one clean row: the ordinary path next to them)
        if mode == "with_masked":
            s[rng.rand(rows, cols) < 0.3] = -np.inf
        truth_l = [np.sort(rng.choice(cols, rng.randint(1, 60), replace=False)).tolist() for _ in range(rows)]
        want, want_topk = native.eval_matrix(s, truth_l, [1, 2, 3, 4, 5], k, want_topk=True)
        got, got_topk, n_exact = eng.eval_scores(_dev(s), _csr(eng, truth_l, cols), [1, 2, 3, 4, 5], k, want_topk=True,
                                                 want_exact_count=True)
        np.testing.assert_array_equal(got_topk.cpu().numpy(), want_topk, err_msg=mode)
        np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32), err_msg=mode)
        assert int(n_exact) >= int(np.isnan(s).any(1).sum()), mode
        if k <= cols:
            np.testing.assert_array_equal(eng.arg_topk(_dev(s), k).cpu().numpy(), native.arg_topk(s, k), err_msg=mode)


def test_eval_users_indirection_and_metric_order(eng):
    """Truth looked up through user ids in a global test CSR; metric order follows the argument."""
    from oracle import native
    rng = np.random.RandomState(4)
    n_users, cols, rows, k = 50, 700, 23, 10
    test_lists = [np.sort(rng.choice(cols, rng.randint(1, 25), replace=False)).tolist()
                  for _ in range(n_users)]
    users = rng.choice(n_users, rows, replace=False).astype(np.int32)
    s = rng.randn(rows, cols).astype(np.float32)
    mids = [4, 1, 5]
    want = native.eval_matrix(s, [test_lists[u] for u in users], mids, k)
    got = eng.eval_scores(_dev(s), _csr(eng, test_lists, cols), mids, k, users=_dev(users))
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_eval_argument_errors(eng):
    s = _dev(np.zeros((2, 8), np.float32))
    truth = _csr(eng, [[1], [2]], 8)
    with pytest.raises(ValueError):
        eng.eval_scores(s, truth, [1], 9)              # top_k > columns
    with pytest.raises(ValueError):
        eng.eval_scores(s, truth, [7], 2)              # unknown metric id
    with pytest.raises(ValueError):
        eng.eval_scores(s, truth, [1], 129, cols=8)    # top_k > columns on the any-K path as well


@pytest.mark.parametrize("rows,cols,k,ties", [(70, 900, 129, False), (33, 700, 300, True), (5, 260, 250, True),
                                              (64, 2000, 512, False)])
def test_any_top_k_matches_the_reference_evaluator(eng, rows, cols, k, ties):
    """evaluate.h:23-50 takes any K: beyond the parallel selection's 128 a thread per row replays
    std::partial_sort_copy (2K of cols, heap tie order) and the metric loops — equal bits to the C++ restatement,
    with ties straddling the cut-off, -inf (masked) entries and truth lists longer than K."""
    from oracle import native
    rng = np.random.RandomState(rows + k)
    s = rng.randn(rows, cols).astype(np.float32)
    if ties:
        s = np.round(s * 4) / 4
    s[rng.rand(rows, cols) < 0.05] = -np.inf
    test_lists = [np.sort(rng.choice(cols, rng.randint(1, min(cols, 2 * k)), replace=False)).tolist() for _ in range(rows)]
    mids = [1, 2, 3, 4, 5]
    want = native.eval_matrix(s, test_lists, mids, k)
    got, topk = eng.eval_scores(_dev(s), _csr(eng, test_lists, cols), mids, k, want_topk=True)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert topk.shape == (rows, k)


def test_mask_train_matches_reference_loop(eng):
    from oracle import native
    rng = np.random.RandomState(8)
    n_users, cols, rows = 40, 333, 17
    train_l = [np.sort(rng.choice(cols, rng.randint(0, 60), replace=False)).tolist()
               for _ in range(n_users)]
    users = rng.choice(n_users, rows, replace=False).astype(np.int32)
    s = rng.randn(rows, cols).astype(np.float32)
    tr = _csr(eng, train_l, cols)
    d = _dev(s)
    eng.mask_train(d, _dev(users), tr)
    want = native.mask_train(s.copy(), users, tr.h_indptr, tr.indices.cpu().numpy())
    np.testing.assert_array_equal(d.cpu().numpy(), want)


@pytest.mark.parametrize("d", [16, 20, 32, 48, 50, 64, 128])
@pytest.mark.parametrize("rows,cols", [(5, 70), (200, 1682), (64, 4097)])
def test_score_gemm_is_the_fmaf_chain_bit_for_bit(eng, d, rows, cols):
    from oracle import native
    rng = np.random.RandomState(d * 7 + rows)
    n_users = rows + 13
    P = (rng.randn(n_users, d) * 0.3).astype(np.float32)
    Q = (rng.randn(cols, d) * 0.3).astype(np.float32)
    users = rng.randint(0, n_users, rows).astype(np.int32)
    gemm = eng.ScoreGemm(_dev(Q), max_rows=rows)
    S = gemm(_dev(P), _dev(users))
    got = S.cpu().numpy()[:, :cols]
    want = native.score_gemm(P, users, Q)
    np.testing.assert_array_equal(got, want)
    assert np.abs(got - P[users].astype(np.float64) @ Q.astype(np.float64).T).max() < 1e-5
    # no gather (users=None) path
    S2 = eng.ScoreGemm(_dev(Q), max_rows=n_users)(_dev(P), None)
    np.testing.assert_array_equal(S2.cpu().numpy()[:, :cols], native.score_gemm(P, None, Q))


def test_full_rank_pipeline_equals_oracle_pipeline(eng):
    """GEMM -> mask -> top-K -> metrics, ml-100k shape and a gowalla-width slice; identical
    rankings and metric values to the CPU statement of the reference driver
    (uni_evaluator.py:132-157)."""
    from neurec_amd.trainer import FullRankEvaluator
    from oracle import native
    rng = np.random.RandomState(12)
    for (U, I, d, n_eval) in [(943, 1682, 64, 943), (600, 40981, 64, 300)]:
        P = (rng.randn(U, d) * 0.1).astype(np.float32)
        Q = (rng.randn(I, d) * 0.1).astype(np.float32)
        train_l = [np.sort(rng.choice(I, rng.randint(1, 80), replace=False)) for _ in range(U)]
        test_l = []
        for u in range(U):
            cand = np.setdiff1d(rng.choice(I, 40, replace=False), train_l[u])
            test_l.append(np.sort(cand[:rng.randint(1, 20)]).tolist())
        users = np.sort(rng.choice(U, n_eval, replace=False)).astype(np.int32)
        tr, te = _csr(eng, [t.tolist() for t in train_l], I), _csr(eng, test_l, I)
        ev = FullRankEvaluator(tr, te, [1, 2, 4, 3, 5], 20, batch_rows=256)
        got = ev.evaluate_factors(_dev(P), _dev(Q), _dev(users), exact_mean=True)
        S = native.score_gemm(P, users, Q)
        native.mask_train(S, users, tr.h_indptr, tr.indices.cpu().numpy())
        per_user = native.eval_matrix(S, [test_l[u] for u in users], [1, 2, 4, 3, 5], 20)
        want = np.mean(per_user, axis=0)
        np.testing.assert_array_equal(got, want)
        fast = ev.evaluate_factors(_dev(P), _dev(Q), _dev(users), exact_mean=False)
        np.testing.assert_allclose(fast, want, rtol=0, atol=1e-7)     # fp64 column sums on device


def test_gowalla_sized_properties(eng):
    """BASELINE size (29,858 x 40,981): properties that need no CPU twin."""
    import torch
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(3)
    U, I, d, K = 29858, 40981, 64, 20
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    deg = rng.randint(8, 60, U)
    indptr = np.zeros(U + 1, np.int64); indptr[1:] = np.cumsum(deg)
    indices = np.concatenate([np.sort(rng.choice(I, n, replace=False)) for n in deg]).astype(np.int32)
    tr = eng.DeviceCSR(indptr, indices, I)
    tptr = np.arange(U + 1, dtype=np.int64)
    te = eng.DeviceCSR(tptr, rng.randint(0, I, U).astype(np.int32), I)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    gemm = eng.ScoreGemm(_dev(Q), 2048)
    S = gemm(_dev(P), users[:2048])
    eng.mask_train(S, users[:2048], tr, cols=I)
    out, topk = eng.eval_scores(S, te, [1, 2, 3, 4, 5], K, users=users[:2048], cols=I, want_topk=True)
    topk_h, S_h = topk.cpu().numpy(), S.cpu().numpy()[:, :I]
    picked = np.take_along_axis(S_h, topk_h.astype(np.int64), axis=1)
    assert np.all(np.diff(picked, axis=1) <= 0)                       # sorted by score
    kth = picked[:, -1]
    assert np.all((S_h > kth[:, None]).sum(1) <= K - 1)               # nothing better was left out
    for r in range(0, 2048, 97):                                       # no train item is ranked
        assert not np.isin(topk_h[r], indices[indptr[r]:indptr[r + 1]]).any()
    o = out.cpu().numpy().reshape(2048, 5, K)
    assert np.all(np.diff(o[:, 1], axis=1) >= 0) and np.all(np.diff(o[:, 4], axis=1) >= 0)  # recall, mrr monotone
    assert np.all((o >= 0) & (o <= 1))
    # whole-population run: every user ranked, result independent of the batch size
    a = FullRankEvaluator(tr, te, [2, 4], K, batch_rows=4096).evaluate_factors(_dev(P), _dev(Q), users)
    b = FullRankEvaluator(tr, te, [2, 4], K, batch_rows=1000).evaluate_factors(_dev(P), _dev(Q), users)
    np.testing.assert_array_equal(a, b)


def test_colsum(eng):
    rng = np.random.RandomState(2)
    m = rng.rand(3001, 100).astype(np.float32)
    got = eng.colsum(_dev(m)).cpu().numpy()
    np.testing.assert_allclose(got, m.astype(np.float64).sum(0), rtol=1e-13)


@pytest.mark.parametrize("d,ties", [(64, False), (50, False), (16, True)])
def test_pruned_evaluation_equals_materialised_scores(d, ties):
    """nrhip_score_tilemax + nrhip_eval_tiles (no score matrix) == score GEMM -> mask -> select:
    identical per-user metric rows, with tie-dependent rows recomputed through the full path."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(d)
    U, I = 700, 5000                                   # 79 tiles of 64 items, last one partial
    P = rng.randn(U, d).astype(np.float32) * 0.3
    Q = rng.randn(I, d).astype(np.float32) * 0.3
    if ties:                                           # coarse values: many exactly equal scores
        P, Q = np.round(P * 2) / 2, np.round(Q * 2) / 2
        Q[100:140] = Q[100]                            # identical items spanning two tiles
    tr = sp.random(U, I, 0.01, random_state=1, format="csr", dtype=np.float32)
    hot = rng.choice(I, 300, replace=False)            # make train items score high for their users
    tr = tr.tolil()
    for u in range(0, U, 3):
        best = np.argsort(-(P[u] @ Q.T))[:rng.randint(1, 30)]
        tr[u, best] = 1.0
    tr = tr.tocsr(); tr.data[:] = 1.0; tr.sort_indices()
    te = sp.random(U, I, 0.004, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
    users = np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    ud = torch.from_numpy(users).cuda()
    full = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=256, pruned=False)
    lean = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=256, pruned=True)
    a = full.evaluate_factors(Pd, Qd, ud, exact_mean=True)
    b = lean.evaluate_factors(Pd, Qd, ud, exact_mean=True)
    np.testing.assert_array_equal(a, b)
    if ties:
        assert lean.n_flagged > 0                      # the tie rows really went through the full path
    else:
        assert lean.n_flagged <= len(users) // 20


@pytest.mark.parametrize("d,U,I,shuffle", [(64, 700, 5000, False), (50, 300, 2500, True), (16, 900, 4133, True),
                                           (128, 200, 1000, False)])
def test_planned_strikes_leave_the_same_tile_maxima_as_the_in_loop_strikes(d, U, I, shuffle):
    """nrhip_score_tilemax without the train lists + nrhip_score_tilemax_fix (the planned (user, tile) pairs
    recomputed with their strikes) == nrhip_score_tilemax with cursors and strikes in the scoring loop, bit for bit —
    for a user subset in any order, in batches, with users whose whole tile is train items, pad columns, d not a
    multiple of 4."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    rng = np.random.RandomState(d + U)
    P = (rng.randn(U, d) * 0.3).astype(np.float32)
    Q = (rng.randn(I, d) * 0.3).astype(np.float32)
    tr = sp.random(U, I, 0.02, random_state=3, format="lil", dtype=np.float32)
    tr[5, 64:96] = 1.0                                  # a whole 32-item tile struck: its maximum is -inf
    tr[7, I - 40:I] = 1.0                               # the last, partial tile
    tr[9, :] = 0.0                                      # a user without train items
    tr = tr.tocsr(); tr.data[:] = 1.0; tr.sort_indices()
    trc = E.DeviceCSR.from_scipy(tr)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    users = np.arange(U, dtype=np.int32)
    if shuffle:
        users = rng.permutation(U)[:U * 2 // 3].astype(np.int32)
    ud = torch.from_numpy(users).cuda()
    gemm = E.ScoreGemm(Qd, 256)
    plan = E.TileStrikePlan(trc, I)
    assert plan.n_pairs <= tr.nnz and plan.n_chunks >= 1
    row_of = torch.full((U,), -1, dtype=torch.int32, device="cuda")
    row_of[ud.long()] = torch.arange(len(users), dtype=torch.int32, device="cuda")
    for b in range(0, len(users), 256):
        u = ud[b:b + 256]
        want = gemm.tile_maxima(Pd, u, trc).clone()
        got = gemm.tile_maxima(Pd, u, trc, plan=plan, row_of=row_of, row_lo=b)
        n_t = (I + 31) // 32
        np.testing.assert_array_equal(got.cpu().numpy()[:, :n_t], want.cpu().numpy()[:, :n_t])
    if not shuffle:                                     # row = user without a lookup table
        want = gemm.tile_maxima(Pd[:256].contiguous(), None, trc).clone()
        got = gemm.tile_maxima(Pd[:256].contiguous(), None, trc, plan=plan)
        n_t = (I + 31) // 32                            # columns beyond the tiles are padding (never written)
        np.testing.assert_array_equal(got.cpu().numpy()[:, :n_t], want.cpu().numpy()[:, :n_t])
        assert np.isneginf(got.cpu().numpy()[5, 2])


def _spread_tables(rng, U, I, d, kind):
    """Factor tables that stress the bf16 expansion: plain gaussians at three scales, and entries whose exponents
    are spread over 2^-20 .. 2^6 (the high / low parts of different coordinates then differ by many binades)."""
    if kind == "wide":
        P = (rng.randn(U, d) * np.exp2(rng.randint(-20, 7, size=(U, d)))).astype(np.float32)
        Q = (rng.randn(I, d) * np.exp2(rng.randint(-20, 7, size=(I, d)))).astype(np.float32)
    elif kind == "tiny":             # entries ~2^-70: every product underflows fp32
        P = (rng.randn(U, d) * 2.0 ** -70).astype(np.float32)
        Q = (rng.randn(I, d) * 2.0 ** -70).astype(np.float32)
    elif kind == "underflow-edge":   # products straddle 2^-126: some normal, some denormal, some flushed
        P = (rng.randn(U, d) * np.exp2(rng.randint(-75, -50, size=(U, d)).astype(np.float64))).astype(np.float32)
        Q = (rng.randn(I, d) * np.exp2(rng.randint(-75, -50, size=(I, d)).astype(np.float64))).astype(np.float32)
    elif kind == "cancel":           # heavy cancellation: u = [v, -v] against items [w, w + tiny]: |u·i| << Σ|u_k i_k|
        h = d // 2
        v, w = rng.randn(U, h), rng.randn(I, h)
        P = np.concatenate([v, -v, np.zeros((U, d - 2 * h))], 1).astype(np.float32)
        Q = np.concatenate([w, w * (1 + 1e-6 * rng.randn(I, h)), rng.randn(I, d - 2 * h)], 1).astype(np.float32)
    elif kind == "norm-spread":      # user norms over 20 binades, a handful of items 2^12 larger than the rest
        P = (rng.randn(U, d) * np.exp2(rng.randint(-10, 11, size=(U, 1)).astype(np.float64))).astype(np.float32)
        Q = rng.randn(I, d).astype(np.float32)
        Q[rng.permutation(I)[:7]] *= 4096.0
    else:
        P = (rng.randn(U, d) * kind).astype(np.float32)
        Q = (rng.randn(I, d) * kind).astype(np.float32)
    return P, Q


@pytest.mark.parametrize("d", [8, 16, 24, 32, 48, 50, 64, 96, 100, 128])
@pytest.mark.parametrize("kind", [0.01, 1.0, 300.0, "wide", "tiny", "underflow-edge", "cancel", "norm-spread"])
def test_bounded_filter_stays_within_its_bound(d, kind):
    """csrc/score_bf16.hip: every approximate tile maximum lies within eps[row] = kappa(d)·||u||·max||i|| of the fp32
    chain's maximum (nrhip_score_tilemax without train lists: exact).  kappa carries a factor 1.5 over the derivation,
    so no error may exceed 2/3 of the bound; measured: 0.07 (d = 64) to 0.30 (d = 8, where Cauchy-Schwarz is tight).  Pad tiles agree (-inf), user subsets and partial batches included."""
    import torch
    from neurec_amd import engine as E
    rng = np.random.RandomState(d * 7 + (11 + len(kind) if isinstance(kind, str) else int(kind * 100)))
    U, I = 333, 4133
    P, Q = _spread_tables(rng, U, I, d, kind)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    users = torch.from_numpy(rng.permutation(U)[:300].astype(np.int32)).cuda()
    gemm, filt = E.ScoreGemm(Qd, 512), E.ScoreFilter(Qd, 512)
    n_t = 2 * ((I + 63) // 64)
    mld = (n_t + 3) // 4 * 4
    exact = torch.empty((300, mld), dtype=torch.float32, device="cuda")
    E.call("nrhip_score_tilemax", E._ptr(Pd), Pd.stride(0), E._ptr(users), 300, I, d, None, None, E._ptr(exact),
           exact.stride(0), E._ptr(gemm.ws), gemm.ws.numel(), E._stream())
    M, eps = filt.tile_maxima(Pd, users)
    a, b, e = exact.cpu().numpy()[:, :n_t], M.cpu().numpy()[:, :n_t], eps.cpu().numpy()
    assert np.array_equal(np.isneginf(a), np.isneginf(b))
    assert np.isneginf(a[:, (I + 31) // 32:]).all() and np.isfinite(a[:, :(I + 31) // 32]).all()
    fin = np.isfinite(a)
    err = np.where(fin, np.abs(np.where(fin, a, 0) - np.where(fin, b, 0)), 0.0)
    un = np.linalg.norm(P[users.cpu().numpy()].astype(np.float64), axis=1)
    imax = np.linalg.norm(Q.astype(np.float64), axis=1).max()
    # eps = the relative bound (rounded up) + the absolute term that covers flushed sub-normal quantities (r05)
    rel, ab = filt.kappa * un * imax, 2.0 ** -110 * d * (1.0 + un + imax)
    assert (e >= rel * (1 - 1e-6)).all() and (e <= (rel + ab) * (1 + 1e-5) + 1e-44).all()
    assert (err <= 0.6 * e[:, None]).all(), "worst error / bound = %.3f" % (err / e[:, None]).max()


@pytest.mark.parametrize("d,clustered,extra", [(64, False, 2), (50, False, 0), (16, True, 2), (32, True, 1), (128, False, 2),
                                               (100, True, 2), (64, "tiny", 2), (48, "underflow-edge", 2)])
def test_bounded_search_ranks_exactly_what_the_fp32_search_ranks(d, clustered, extra):
    """FullRankEvaluator(search='bf16') == search='fp32' == the materialised path, per-user metric rows bit for bit.
    `clustered`: items that are tiny perturbations of each other, spread over many tiles — the gaps between the best
    scores fall below the bound, the certificate fails, the rows are redone from fp32 rows (n_flagged > 0)."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(d + 5)
    U, I = 600, 6000
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    if isinstance(clustered, str):
        # magnitudes a relative bound cannot certify: scores in / below the sub-normal range — the
        # absolute term of eps fails the certificate and the rows are ranked from fp32 rows
        P, Q = _spread_tables(rng, U, I, d, clustered)
    elif clustered:
        base = Q[:60].copy()
        for c in range(100):                               # 100 near-copies of 60 items, scattered over the tiles
            Q[c * 60:(c + 1) * 60] = base * (1.0 + rng.randn(60, 1).astype(np.float32) * 1e-7)
        Q = Q[rng.permutation(I)]
    tr = sp.random(U, I, 0.01, random_state=1, format="lil", dtype=np.float32)
    for u in range(0, U, 3):
        tr[u, np.argsort(-(P[u] @ Q.T))[:rng.randint(1, 30)]] = 1.0
    tr = tr.tocsr(); tr.data[:] = 1.0; tr.sort_indices()
    te = sp.random(U, I, 0.004, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
    users = np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    ud = torch.from_numpy(users).cuda()
    full = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=256, pruned=False)
    exact = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=256, search="fp32")
    fast = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=256, search="bf16", extra_tiles=extra)
    a = full.evaluate_factors(Pd, Qd, ud, exact_mean=True)
    b = exact.evaluate_factors(Pd, Qd, ud, exact_mean=True)
    c = fast.evaluate_factors(Pd, Qd, ud, exact_mean=True)
    assert exact.search_used == "fp32" and fast.search_used == "bf16"
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)
    if isinstance(clustered, str):
        assert fast.n_flagged > len(users) // 2              # uncertifiable magnitudes: the fp32 path decides
    elif clustered:
        assert fast.n_flagged > 0
    else:
        assert fast.n_flagged <= len(users) // 20
    # the float64 means (one device->host copy, flagged rows redone on demand) agree as well
    np.testing.assert_array_equal(exact.evaluate_factors(Pd, Qd, ud), fast.evaluate_factors(Pd, Qd, ud))


def test_bounded_search_falls_back_where_it_is_not_built():
    """A user list with repeats has no strike plan and takes the fp32 search with in-loop strikes; an unknown search
    name is refused; the filter itself is built up to 128 columns."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(3)
    U, I, d = 100, 3000, 96
    P, Q = (rng.randn(U, d) * 0.1).astype(np.float32), (rng.randn(I, d) * 0.1).astype(np.float32)
    tr = sp.random(U, I, 0.01, random_state=1, format="csr", dtype=np.float32); tr.data[:] = 1.0
    te = sp.random(U, I, 0.01, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    u = np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)
    users = torch.from_numpy(np.concatenate([u, u[:7]])).cuda()               # seven users twice
    ev = FullRankEvaluator(trc, tec, [1, 3], 10, batch_rows=64)
    ref = FullRankEvaluator(trc, tec, [1, 3], 10, batch_rows=64, pruned=False)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    np.testing.assert_array_equal(ev.evaluate_factors(Pd, Qd, users, exact_mean=True),
                                  ref.evaluate_factors(Pd, Qd, users, exact_mean=True))
    assert ev.search_used == "fp32"
    ev.evaluate_factors(Pd, Qd, torch.from_numpy(u).cuda())
    assert ev.search_used == "int8"                                           # (the default search, built up to 128 columns)
    assert not E.ScoreFilter.supports(129) and E.ScoreFilter.supports(128)
    with pytest.raises(ValueError, match="search"):
        FullRankEvaluator(trc, tec, [1], 10, search="fp16")
    with pytest.raises(NotImplementedError):
        E.ScoreFilter(torch.zeros((I, 160), device="cuda"), 64)


@pytest.mark.parametrize("d,U,I", [(64, 700, 5000), (50, 300, 2500), (16, 257, 4133), (128, 130, 1000), (24, 90, 500)])
def test_tile_grouped_rescoring_equals_the_per_row_kernel(d, U, I):
    """nrhip_eval_tiles_bounded rescoring bucketed by tile (32 users of one tile per wave, fp32 MFMA chain) ==
    nrhip_eval_tiles' one-wave-per-user fmaf chain: metric rows and flags identical — popular tiles that every user
    picks (one bucket holding all rows), the partial last tile, users without train items."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    rng = np.random.RandomState(d + U)
    P = (rng.randn(U, d) * 0.3).astype(np.float32)
    Q = (rng.randn(I, d) * 0.3).astype(np.float32)
    Q[40:60] += 3.0 * P.mean(0) / max(np.linalg.norm(P.mean(0)), 1e-6)       # items most users rank high
    P += 0.5 * P.mean(0)
    tr = sp.random(U, I, 0.02, random_state=3, format="lil", dtype=np.float32)
    tr[7, I - 40:I] = 1.0
    tr[9, :] = 0.0
    tr = tr.tocsr(); tr.data[:] = 1.0; tr.sort_indices()
    te = sp.random(U, I, 0.01, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    users = torch.from_numpy(rng.permutation(U).astype(np.int32)).cuda()
    gemm = E.ScoreGemm(Qd, U)
    M = gemm.tile_maxima(Pd, users, trc)
    k = 10
    outs = []
    for grouped in (False, True):
        out = torch.zeros((U, 3 * k), dtype=torch.float32, device="cuda")
        flags = torch.full((U,), -1, dtype=torch.int32, device="cuda")
        E.eval_tiles(M, Pd, gemm, users, trc, tec, [1, 3, 5], k, out, flags, grouped=grouped)
        outs.append((out.cpu().numpy(), flags.cpu().numpy()))
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    keep = outs[0][1] == 0
    np.testing.assert_array_equal(outs[0][0][keep], outs[1][0][keep])
    assert keep.sum() > U // 2


def test_ties_that_cannot_change_a_metric_do_not_send_a_row_to_the_replay():
    """Two equal scores at neighbouring ranks matter only if one is a test item and the other is not, or if the pair
    straddles the cut: every metric is a function of the hit / miss sequence alone.  Duplicate item rows make exact
    ties for every user; the pruned evaluation still equals the materialised one (which replays the reference's heap
    for every tie) and flags only the rows whose ties could matter."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(11)
    U, I, d = 500, 4000, 32
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    Q[1000:2000] = Q[:1000]                             # every item of the first thousand has an exact duplicate
    tr = sp.random(U, I, 0.005, random_state=1, format="csr", dtype=np.float32); tr.data[:] = 1.0
    te = sp.random(U, I, 0.004, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
    users = np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    Pd, Qd, ud = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda(), torch.from_numpy(users).cuda()
    full = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=256, pruned=False)
    a = full.evaluate_factors(Pd, Qd, ud, exact_mean=True)
    for search in ("bf16", "fp32"):
        lean = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=256, search=search)
        b = lean.evaluate_factors(Pd, Qd, ud, exact_mean=True)
        np.testing.assert_array_equal(a, b)
        # nearly every user has a tied pair among its 20 best (half of the top items are duplicated); few of those
        # pairs involve a test item or the cut
        assert 0 < lean.n_flagged < len(users) // 3, lean.n_flagged


@pytest.mark.parametrize("search", ["bf16", "fp32"])
def test_the_native_batch_loop_equals_the_python_batch_loop(search):
    """nrhip_eval_pruned (the batch loop, the column sums and the flagged-row count in one call) == the same entry points
    issued from Python batch by batch: per-user rows, flags, the float64 means; several batches, a short last one,
    a shuffled user subset, a second evaluation with other tables through the same evaluator."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(21)
    U, I, d = 700, 3000, 48
    tr = sp.random(U, I, 0.01, random_state=1, format="csr", dtype=np.float32); tr.data[:] = 1.0
    te = sp.random(U, I, 0.005, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
    users = rng.permutation(np.flatnonzero(np.diff(te.indptr) > 0))[:611].astype(np.int32)
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    ud = torch.from_numpy(users).cuda()
    a = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, search=search)
    b = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, search=search)
    a.native_loop, b.native_loop = True, False
    for seed in (0, 1):
        r2 = np.random.RandomState(seed)
        Pd = torch.from_numpy((r2.randn(U, d) * 0.1).astype(np.float32)).cuda()
        Qd = torch.from_numpy((r2.randn(I, d) * 0.1).astype(np.float32)).cuda()
        np.testing.assert_array_equal(a.evaluate_factors(Pd, Qd, ud), b.evaluate_factors(Pd, Qd, ud))
        assert a.n_flagged == b.n_flagged
        np.testing.assert_array_equal(a.evaluate_factors(Pd, Qd, ud, exact_mean=True),
                                      b.evaluate_factors(Pd, Qd, ud, exact_mean=True))
    assert a._native is not None and getattr(b, "_native", None) is None


def test_nan_user_rows_do_not_disturb_the_other_users():
    """User factor rows full of NaN (a diverged model) leave garbage tile ids behind the selection; the tile buckets
    take one pair per row, anything beyond is dropped and the row flagged — no fault, and the other users' rows are
    what the materialised path computes for them."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(5)
    U, I, d = 400, 3000, 32
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    bad = rng.choice(U, 40, replace=False)
    P[bad] = np.nan
    tr = sp.random(U, I, 0.01, random_state=1, format="csr", dtype=np.float32); tr.data[:] = 1.0
    te = sp.random(U, I, 0.01, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    everyone = np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)
    good = np.setdiff1d(everyone, bad).astype(np.int32)
    lean = FullRankEvaluator(trc, tec, [1, 3, 5], 10, batch_rows=128)
    full = FullRankEvaluator(trc, tec, [1, 3, 5], 10, batch_rows=128, pruned=False)
    lean.evaluate_factors(Pd, Qd, torch.from_numpy(everyone).cuda())          # must not fault
    torch.cuda.synchronize()
    g = torch.from_numpy(good).cuda()
    np.testing.assert_array_equal(lean.evaluate_factors(Pd, Qd, g, exact_mean=True),
                                  full.evaluate_factors(Pd, Qd, g, exact_mean=True))


@pytest.mark.parametrize("shape", [(300, 1000, 0.02), (1, 40, 0.5), (64, 33, 0.3), (500, 70001, 0.0005)])
def test_native_strike_plan_lists_every_user_tile_pair_once(shape):
    """engine.TileStrikePlan (nrhip_tile_strike_plan, r05: no torch ops): the (user, tile) pairs, their item bits,
    the per-tile pointers and the chunk table against a numpy construction — as SETS per tile (the order inside a
    tile is unspecified), duplicate items in a hand-made CSR included."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    U, I, dens = shape
    tr = sp.random(U, I, dens, random_state=U + I, format="csr", dtype=np.float32)
    tr.sort_indices()
    indptr, indices = tr.indptr.astype(np.int64), tr.indices.astype(np.int32)
    if len(indices) > 3:                                   # a repeated item (a CSR assembled by hand may hold one)
        r = int(np.flatnonzero(np.diff(indptr) >= 2)[0]) if (np.diff(indptr) >= 2).any() else None
        if r is not None:
            indices[indptr[r] + 1] = indices[indptr[r]]
    csr = E.DeviceCSR(indptr, indices, I)
    plan = E.TileStrikePlan(csr, I)
    n_tiles = 2 * ((I + 63) // 64)
    want = {}
    for u in range(U):
        for i in indices[indptr[u]:indptr[u + 1]]:
            want[(int(i) >> 5, u)] = want.get((int(i) >> 5, u), 0) | (1 << (int(i) & 31))
    assert plan.n_pairs == len(want)
    tp = plan.tile_ptr.cpu().numpy()
    user, mask = plan.user.cpu().numpy(), plan.mask.cpu().numpy().view(np.uint32)
    assert tp[0] == 0 and tp[-1] == len(want) and len(tp) == n_tiles + 1 and (np.diff(tp) >= 0).all()
    got = {}
    for t in range(n_tiles):
        for k in range(tp[t], tp[t + 1]):
            assert (t, int(user[k])) not in got
            got[(t, int(user[k]))] = int(mask[k])
    assert got == want
    ct, cb = plan.chunk_tile.cpu().numpy()[:plan.n_chunks], plan.chunk_begin.cpu().numpy()[:plan.n_chunks]
    assert plan.n_chunks == int(((np.diff(tp) + 31) // 32).sum())
    covered = np.zeros(len(want), bool)
    for t, b in zip(ct, cb):
        e = min(b + 32, tp[t + 1])
        assert tp[t] <= b < e and not covered[b:e].any()
        covered[b:e] = True
    assert covered.all()


def test_strike_plan_refuses_a_train_matrix_it_cannot_plan():
    """the native plan build takes the first entry of a (user, tile) run as the pair's head — a row whose
    items do not ascend would give one pair two heads with partial masks (two different maxima for one M slot, no
    flag) — and indexes its histograms with the item ids.  Both preconditions are checked once, at plan build: an
    unsorted row or an item id >= cols raises instead; a repeated item (neighbours in an ascending row) is fine."""
    import torch
    from neurec_amd import engine as E
    I = 500
    ptr = np.asarray([0, 3, 6], np.int64)
    ok = E.DeviceCSR(ptr, np.asarray([5, 40, 41, 7, 7, 300], np.int32), I)            # ascending, one repeat
    assert E.TileStrikePlan(ok, I).n_pairs == 4                                        # user 0: tiles 0, 1 (40 and 41 share it); user 1: tiles 0 (7 twice), 9
    with pytest.raises(ValueError, match="ascending"):
        E.TileStrikePlan(E.DeviceCSR(ptr, np.asarray([40, 5, 41, 7, 8, 300], np.int32), I), I)
    with pytest.raises(ValueError, match="outside"):
        E.TileStrikePlan(E.DeviceCSR(ptr, np.asarray([5, 40, 41, 7, 8, 500], np.int32), I), I)
    # the evaluator on an unsorted hand-made CSR says so as well (it builds the plan on first use)
    from neurec_amd.trainer import FullRankEvaluator
    bad = E.DeviceCSR(ptr, np.asarray([40, 5, 41, 7, 8, 300], np.int32), I)
    te = E.DeviceCSR(ptr, np.asarray([1, 2, 3, 4, 5, 6], np.int32), I)
    P, Q = torch.randn(2, 16, device="cuda"), torch.randn(I, 16, device="cuda")
    with pytest.raises(ValueError, match="ascending"):
        FullRankEvaluator(bad, te, [1], 5, batch_rows=64).evaluate_factors(P, Q, torch.arange(2, dtype=torch.int32, device="cuda"))
