"""csrc/score_i8.hip: the bounded search on the int8 matrix cores (d <= 128; beyond 64 columns a tile's k range
runs in two halves).  Same contract as the bf16 filter: its
tile maxima are never ranked, every one carries a bound on its distance from the fp32 chain's value, and
FullRankEvaluator(search="int8") returns exactly the rows the fp32 search and the materialised path return
(evaluator/backend/cpp/src/evaluate.h:23-50 stays the definition; MF.py:120-122 the scores)."""
import numpy as np
import pytest

from test_eval_gpu import _spread_tables

pytestmark = pytest.mark.gpu


def _fixed_point_bound(P, Q, d):
    """score_i8.hip's derivation restated in float64 numpy from the same integers the split kernels form: the
    user-side part of eps (without the chain's own rounding term) and the per-(user, 32-item tile) term that the
    filter adds INTO its maxima (0.525 su sI max_{i in tile} Σ|qi|)."""
    aI = np.abs(Q).max()
    sI = np.float32(aI) / np.float32(16256)
    qi = np.clip(np.rint((Q * (np.float32(16256) / np.float32(aI))).astype(np.float32)), -16256, 16256)
    au = np.abs(P).max(1)
    su = (au / np.float32(16256)).astype(np.float32)
    qu = np.clip(np.rint((P * (np.float32(16256) / au)[:, None]).astype(np.float32)), -16256, 16256)
    lu = qu - 128 * np.floor((qu + 64) / 128)
    susi = su.astype(np.float64) * float(sI)
    user = susi * (0.525 * np.abs(qu).sum(1) + 0.27 * d + 64 * np.abs(lu).sum(1))
    I = Q.shape[0]
    nt = (I + 31) // 32
    q1 = np.zeros(nt * 32)
    q1[:I] = np.abs(qi).sum(1)
    tile = 0.525 * susi[:, None] * q1.reshape(nt, 32).max(1)[None, :]
    return user, tile


@pytest.mark.parametrize("d", [8, 16, 24, 32, 48, 50, 64, 96, 100, 128])
@pytest.mark.parametrize("kind", [0.01, 1.0, 300.0, "wide", "cancel", "norm-spread"])
def test_int8_filter_stays_within_its_derived_bound(d, kind):
    """The filter's tile maxima are UPPER-bound maxima: fp32 chain maximum of the tile (nrhip_score_tilemax without
    train lists: exact) <= M[row][tile] + eps[row] — what the certificate needs — and they are no looser than the
    derivation allows: M - chain maximum <= eps[row] + 2 x the tile term.  Gaussians at three scales, exponents spread
    over 26 binades, heavy cancellation, norms spread over 20 binades.  The integer accumulators are exact, so the
    bound is the derivation itself (no safety factor over a measured model); eps is checked against its formula."""
    import torch
    from neurec_amd import engine as E
    rng = np.random.RandomState(d * 7 + (11 + len(kind) if isinstance(kind, str) else int(kind * 100)))
    U, I = 333, 4133
    P, Q = _spread_tables(rng, U, I, d, kind)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    users_h = rng.permutation(U)[:300].astype(np.int32)
    users = torch.from_numpy(users_h).cuda()
    gemm, filt = E.ScoreGemm(Qd, 512), E.ScoreFilter(Qd, 512, "int8")
    n_t = 2 * ((I + 63) // 64)
    mld = (n_t + 3) // 4 * 4
    exact = torch.empty((300, mld), dtype=torch.float32, device="cuda")
    E.call("nrhip_score_tilemax", E._ptr(Pd), Pd.stride(0), E._ptr(users), 300, I, d, None, None, E._ptr(exact),
           exact.stride(0), E._ptr(gemm.ws), gemm.ws.numel(), E._stream())
    M, eps = filt.tile_maxima(Pd, users)
    a, b, e = exact.cpu().numpy()[:, :n_t], M.cpu().numpy()[:, :n_t], eps.cpu().numpy().astype(np.float64)
    assert np.array_equal(np.isneginf(a), np.isneginf(b))
    n_real = (I + 31) // 32
    assert np.isneginf(a[:, n_real:]).all() and np.isfinite(a[:, :n_real]).all()
    assert np.isfinite(e).all()
    a64, b64 = a[:, :n_real].astype(np.float64), b[:, :n_real].astype(np.float64)
    user, tile = _fixed_point_bound(P[users_h], Q, d)
    over = (a64 - b64) / e[:, None]                          # chain maximum above the stored maximum, in bounds
    assert (over <= 1.0).all(), "chain maximum exceeds M + eps: worst (chain - M) / eps = %.3f" % over.max()
    slack = b64 - a64 - (e[:, None] + 2.0 * tile * (1 + 1e-5))
    assert (slack <= 0).all(), "M looser than eps + 2 x tile term by %.3e" % slack.max()
    # eps is the one the header derives: user-side quantisation part + 1.5 d 2^-24 ||u|| max||i|| (+ the absolute term)
    un = np.linalg.norm(P[users_h].astype(np.float64), axis=1)
    imax = np.linalg.norm(Q.astype(np.float64), axis=1).max()
    want = user + 1.5 * d * 2.0 ** -24 * un * imax
    ab = 2.0 ** -110 * d * (1.0 + un + imax)
    assert (e >= want * (1 - 1e-5)).all() and (e <= (want + ab) * (1 + 1e-5) + 1e-44).all()
    # and a useful one: bound + the largest tile term stay a small multiple of the bf16 form's bound
    kappa = E.ScoreFilter(Qd, 512).kappa
    assert (e + tile.max(1) <= 12.0 * kappa * un * imax + ab).all()


@pytest.mark.parametrize("kind", ["tiny", "underflow-edge", "nan-user", "inf-item", "zero-user"])
def test_int8_filter_refuses_to_bound_what_fixed_point_cannot_hold(kind):
    """Magnitudes outside 2^-40 .. 2^40 and non-finite entries: eps = NaN (every certificate fails, the caller's fp32
    path ranks the row); rows of zeros are exact with a bound of (almost) zero."""
    import torch
    from neurec_amd import engine as E
    rng = np.random.RandomState(17)
    U, I, d = 70, 1000, 32
    if kind in ("tiny", "underflow-edge"):
        P, Q = _spread_tables(rng, U, I, d, kind)
    else:
        P, Q = (rng.randn(U, d) * 0.1).astype(np.float32), (rng.randn(I, d) * 0.1).astype(np.float32)
    if kind == "nan-user":
        P[3, 5] = np.nan
        P[9, 0] = np.inf
    if kind == "inf-item":
        Q[77, 1] = np.inf
    if kind == "zero-user":
        P[4] = 0.0
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    M, eps = E.ScoreFilter(Qd, 128, "int8").tile_maxima(Pd, None)
    e, m = eps.cpu().numpy(), M.cpu().numpy()
    if kind in ("tiny", "underflow-edge", "inf-item"):
        assert np.isnan(e).all()
    elif kind == "nan-user":
        assert np.isnan(e[[3, 9]]).all() and np.isfinite(np.delete(e, [3, 9])).all()
    else:
        assert np.isfinite(e).all() and e[4] < 1e-30
        assert (np.abs(m[4, :(I + 31) // 32]) < 1e-30).all()
    assert not np.isnan(m[:, :(I + 31) // 32]).any()


@pytest.mark.parametrize("d,clustered,extra", [(64, False, 2), (50, False, 0), (16, True, 2), (32, True, 1), (8, False, 2),
                                               (64, "tiny", 2), (48, "underflow-edge", 2), (128, False, 2), (96, False, 0),
                                               (128, True, 2), (100, "tiny", 2)])
def test_int8_search_ranks_exactly_what_the_fp32_search_ranks(d, clustered, extra):
    """FullRankEvaluator(search='int8') == search='fp32' == the materialised path, per-user metric rows bit for bit;
    near-duplicate items (the certificate fails) and magnitudes the fixed point cannot hold (eps = NaN) take the fp32
    path and come out the same."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(d + 5)
    U, I = 600, 6000
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    if isinstance(clustered, str):
        P, Q = _spread_tables(rng, U, I, d, clustered)
    elif clustered:
        base = Q[:60].copy()
        for c in range(100):
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
    fast = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=256, search="int8", extra_tiles=extra)
    a = full.evaluate_factors(Pd, Qd, ud, exact_mean=True)
    b = exact.evaluate_factors(Pd, Qd, ud, exact_mean=True)
    c = fast.evaluate_factors(Pd, Qd, ud, exact_mean=True)
    assert exact.search_used == "fp32" and fast.search_used == "int8"
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)
    if isinstance(clustered, str):
        assert fast.n_flagged == len(users)                  # nothing is certified on a NaN bound
    elif clustered:
        assert fast.n_flagged > 0
    else:
        assert fast.n_flagged <= len(users) // 20
    np.testing.assert_array_equal(exact.evaluate_factors(Pd, Qd, ud), fast.evaluate_factors(Pd, Qd, ud))


def test_int8_search_native_loop_python_loop_and_width_fallback():
    """The one-call batch loop (nrhip_eval_pruned, use_filter = 2) equals the Python batch loop; 65 .. 128 columns run
    the two-half kernel; the int8 entry points refuse d > 128 by name."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(9)
    U, I, d = 500, 5000, 64
    P, Q = (rng.randn(U, d) * 0.1).astype(np.float32), (rng.randn(I, d) * 0.1).astype(np.float32)
    tr = sp.random(U, I, 0.01, random_state=1, format="csr", dtype=np.float32); tr.data[:] = 1.0
    te = sp.random(U, I, 0.005, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    users = torch.from_numpy(np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)).cuda()
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    a = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, search="int8")
    b = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, search="int8")
    a.native_loop, b.native_loop = True, False
    ref = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, pruned=False)
    ra = a.evaluate_factors(Pd, Qd, users, per_user=True)
    rb = b.evaluate_factors(Pd, Qd, users, per_user=True)
    rr = ref.evaluate_factors(Pd, Qd, users, per_user=True)
    assert a.search_used == "int8" and b.search_used == "int8"
    np.testing.assert_array_equal(np.asarray(ra), np.asarray(rb))
    np.testing.assert_array_equal(np.asarray(ra), np.asarray(rr))
    # tables of 65 .. 128 columns: the same search, a tile's k range in two halves (r06; r05 fell to the bf16 form)
    P2, Q2 = (rng.randn(U, 96) * 0.1).astype(np.float32), (rng.randn(I, 96) * 0.1).astype(np.float32)
    w = FullRankEvaluator(trc, tec, [1, 3], 10, batch_rows=128, search="int8")
    rw = w.evaluate_factors(torch.from_numpy(P2).cuda(), torch.from_numpy(Q2).cuda(), users, per_user=True)
    assert w.search_used == "int8"
    rr = FullRankEvaluator(trc, tec, [1, 3], 10, batch_rows=128, pruned=False).evaluate_factors(
        torch.from_numpy(P2).cuda(), torch.from_numpy(Q2).cuda(), users, per_user=True)
    np.testing.assert_array_equal(np.asarray(rw), np.asarray(rr))
    assert E.ScoreFilter.supports(128, "int8") and not E.ScoreFilter.supports(129, "int8")
    with pytest.raises(NotImplementedError):
        E.ScoreFilter(torch.zeros((I, 160), device="cuda"), 64, "int8")


def test_an_int8_evaluation_that_redid_rows_pauses_int8_for_the_next_ones():
    """The int8 bound is a few times the bf16 form's: on tables whose best scores crowd together it leaves rows
    uncertified, and every such row costs a full fp32 row.  An evaluator that had to redo rows under int8 takes the
    next `int8_retry` evaluations through the bf16 filter and then tries int8 again; results are the same throughout."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(23)
    U, I, d = 300, 6000, 32
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    base = Q[:60].copy()
    for c in range(100):                                    # near-duplicate items over many tiles: certificates fail
        Q[c * 60:(c + 1) * 60] = base * (1.0 + rng.randn(60, 1).astype(np.float32) * 1e-7)
    Q = Q[rng.permutation(I)]
    tr = sp.random(U, I, 0.01, random_state=1, format="csr", dtype=np.float32); tr.data[:] = 1.0
    te = sp.random(U, I, 0.005, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    users = torch.from_numpy(np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)).cuda()
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    ev = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, search="int8")
    ev.int8_retry = 2
    ref = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, pruned=False).evaluate_factors(Pd, Qd, users)
    used = []
    for _ in range(5):
        np.testing.assert_array_equal(ev.evaluate_factors(Pd, Qd, users), ref)
        used.append(ev.search_used)
    assert used == ["int8", "bf16", "bf16", "int8", "bf16"]
    # tables it certifies completely keep int8
    Q2 = (rng.randn(I, d) * 0.1).astype(np.float32)
    ev2 = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, search="int8")
    clean = True
    for _ in range(3):
        ev2.evaluate_factors(Pd, torch.from_numpy(Q2).cuda(), users)
        assert ev2.search_used == ("int8" if clean else "bf16")
        clean = clean and ev2.n_flagged == 0
    assert clean                                            # gaussian tables: gaps of tens of bounds


def test_rows_flagged_for_ties_do_not_cost_the_int8_search():
    """exact duplicates among a user's best items flag the row for its TIE (bit 0: redone from a full row so
    that the reference's heap order decides) — that says nothing about the int8 bound, whose certificate (bit 1)
    passed, so the next evaluation still searches in int8."""
    import torch
    import scipy.sparse as sp
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    rng = np.random.RandomState(31)
    U, I, d = 300, 6000, 32
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    Q[:20] *= 5.0                                           # twenty long items lead most rankings ...
    Q[3000:3020] = Q[:20]                                   # ... and each exists twice, in another tile
    tr = sp.random(U, I, 0.01, random_state=1, format="csr", dtype=np.float32); tr.data[:] = 1.0
    te = sp.random(U, I, 0.005, random_state=2, format="csr", dtype=np.float32)
    te = te - te.multiply(tr)
    tr, te = tr.tolil(), te.tolil()
    for u in range(0, 120):                                 # one copy of a pair is a test item, the other is not:
        tr[u, 5] = tr[u, 3005] = te[u, 3005] = 0            # where the pair makes the top 20 its order decides metrics
        te[u, 5] = 1.0
    tr, te = tr.tocsr(), te.tocsr()
    tr.eliminate_zeros(); te.eliminate_zeros(); tr.sort_indices(); te.sort_indices()
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    users = torch.from_numpy(np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)).cuda()
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    ev = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, search="int8")
    ref = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, pruned=False).evaluate_factors(Pd, Qd, users)
    for _ in range(3):
        np.testing.assert_array_equal(ev.evaluate_factors(Pd, Qd, users), ref)
        assert ev.search_used == "int8" and ev.n_flagged > 0 and ev.n_uncertified == 0
