"""The int8 search's bound (csrc/score_i8.hip, header derivation) checked on the host: the split and the integer
products are emulated in numpy exactly as the kernels form them (same float32 operations for the scales and the
quotients, int64 for the products), and the one-sided contract the certificate reads —
    exact dot(u, i) + |chain - exact| <= M[u][tile(i)] + eps[u]
— is verified over random, heavy-tailed and constructed worst-case tables.  No GPU: this pins the DERIVATION; that the
kernels compute these integers is tests/test_filter_i8_gpu.py's part."""
import numpy as np
import pytest

QMAX = 16256


def _split(X, per_row):
    """q (int64), the float32 scale(s) — split_rows_i8_kernel's arithmetic"""
    X = X.astype(np.float32)
    amax = np.abs(X).max(1) if per_row else np.full(X.shape[0], np.abs(X).max(), np.float32)
    amax = amax.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(amax > 0, np.float32(QMAX) / amax, np.float32(0)).astype(np.float32)
        scale = (amax / np.float32(QMAX)).astype(np.float32)
    q = np.rint((X * inv[:, None]).astype(np.float32)).astype(np.int64)
    q = np.clip(q, -QMAX, QMAX)
    h = (q + 64) >> 7
    l = q - (h << 7)
    assert np.abs(h).max() <= 127 and l.min() >= -64 and l.max() <= 63
    return q, h, l, scale


def _filter(P, Q):
    """(M, eps): the stored upper-bound maxima per (user, 32-item tile) and the user-side bound, as the kernels form them"""
    d = P.shape[1]
    qu, hu, lu, su = _split(P, True)
    qi, hi, li, sI = _split(Q, False)
    sI = sI[0]
    V = 128 * (hu @ hi.T) + hu @ li.T + lu @ hi.T                       # int64: what the int32 accumulators hold
    assert np.abs(V).max() < 2 ** 31
    I = Q.shape[0]
    nt = (I + 31) // 32
    Vp = np.full((P.shape[0], nt * 32), np.iinfo(np.int64).min)
    Vp[:, :I] = V
    Vmax = Vp.reshape(-1, nt, 32).max(2)
    q1 = np.zeros(nt * 32)
    q1[:I] = np.abs(qi).sum(1)
    qblk = q1.reshape(nt, 32).max(1).astype(np.float32)
    susi_up = np.nextafter((su * sI).astype(np.float32), np.float32(np.inf))
    cu = (su * sI * np.float32(128)).astype(np.float32)
    tu = np.nextafter((np.float32(0.525) * susi_up).astype(np.float32), np.float32(np.inf))
    M = (qblk[None, :].astype(np.float64) * tu[:, None] + (Vmax.astype(np.float32) * cu[:, None]).astype(np.float32))
    M = M.astype(np.float32)                                            # (the fmaf rounds once: this rounds twice, 1 ulp)
    Q1, L1 = np.abs(qu).sum(1), np.abs(lu).sum(1)
    un = np.linalg.norm(P.astype(np.float64), axis=1)
    imax = np.linalg.norm(Q.astype(np.float64), axis=1).max()
    eps = susi_up.astype(np.float64) * (0.525 * Q1 + 0.27 * d + 64.0 * L1) + 1.5 * d * 2.0 ** -24 * un * imax
    return M.astype(np.float64), eps, Vmax


def _check(P, Q):
    P, Q = P.astype(np.float32), Q.astype(np.float32)
    M, eps, _ = _filter(P, Q)
    d, I = P.shape[1], Q.shape[0]
    exact = P.astype(np.float64) @ Q.astype(np.float64).T
    # the fp32 chain's own distance from the exact dot: gamma_d * sum |u_k i_k| (sequential fmaf, d roundings)
    chain_dev = d * 2.0 ** -24 * (np.abs(P).astype(np.float64) @ np.abs(Q).astype(np.float64).T) * (1 + 1e-6)
    nt = (I + 31) // 32
    up = np.full((P.shape[0], nt * 32), -np.inf)
    up[:, :I] = exact + chain_dev                                       # the largest value the chain can take
    worst = up.reshape(-1, nt, 32).max(2)
    slack = M * (1 + 2.0 ** -22) + eps[:, None] - worst                 # (2 ulp: the emulation's double rounding of M)
    assert (slack >= 0).all(), "chain can exceed M + eps by %.3e (bound %.3e)" % (-slack.min(), eps.max())
    return float((worst - M).max() / eps.max())


@pytest.mark.parametrize("d", [8, 16, 32, 50, 64])
@pytest.mark.parametrize("kind", ["gauss", "heavy", "spread", "one-hot", "half-steps", "all-max", "anti"])
def test_fixed_point_bound_holds_for_the_integers_the_kernels_form(d, kind):
    rng = np.random.RandomState(d * 13 + len(kind))
    U, I = 64, 700
    if kind == "gauss":
        P, Q = rng.randn(U, d) * 0.05, rng.randn(I, d) * 0.05
    elif kind == "heavy":                # item norms over two decades, user entries log-normal
        P = rng.randn(U, d) * np.exp(rng.randn(U, d) * 0.8) * 0.05
        Q = rng.randn(I, d) * np.exp(rng.randn(I, 1) * 1.2) * 0.02
    elif kind == "spread":               # exponents over 26 binades: most entries quantise to 0
        P = rng.randn(U, d) * np.exp2(rng.randint(-20, 7, (U, d)))
        Q = rng.randn(I, d) * np.exp2(rng.randint(-20, 7, (I, d)))
    elif kind == "one-hot":              # one dominant coordinate per row: the scale belongs to it, the rest is noise
        P, Q = rng.randn(U, d) * 1e-3, rng.randn(I, d) * 1e-3
        P[np.arange(U), rng.randint(0, d, U)] = rng.choice([-3.0, 3.0], U)
        Q[np.arange(I), rng.randint(0, d, I)] = rng.choice([-2.0, 2.0], I)
    elif kind == "half-steps":           # every entry half a quantum off a grid point: |du|, |di| at their largest
        ku, ki = rng.randint(-QMAX + 1, QMAX - 1, (U, d)), rng.randint(-QMAX + 1, QMAX - 1, (I, d))
        P = (ku + 0.5) / QMAX
        Q = (ki + 0.5) / QMAX
        P[:, 0], Q[0, 0] = 1.0, 1.0      # pins the scales at 1 / 16256
    elif kind == "all-max":              # |q| = 16256 everywhere: the largest integers the accumulators see
        P, Q = rng.choice([-1.0, 1.0], (U, d)), rng.choice([-1.0, 1.0], (I, d))
    else:                                # "anti": l planes at -64 / +63 with aligned signs: sum lu li as large as it gets
        hu_, hi_ = rng.randint(-100, 100, (U, d)), rng.randint(-100, 100, (I, d))
        P = (128 * hu_ + rng.choice([-64, 63], (U, d))) / QMAX
        Q = (128 * hi_ + rng.choice([-64, 63], (I, d))) / QMAX
        P[:, 0], Q[0, 0] = 1.0, 1.0
    worst = _check(P, Q)
    assert worst <= 1.0 + 1e-9


def test_the_dropped_low_product_is_what_its_term_says():
    """|sum_k lu li| <= 64 sum|lu| is attained (up to the l range's asymmetry) when every li = -64 and lu keeps one sign:
    the 64 Lu1 term of the bound is not slack that could be dropped."""
    d = 64
    lu = np.full(d, -64)
    li = np.full(d, -64)
    assert abs(int((lu * li).sum())) == 64 * int(np.abs(lu).sum())


def test_zero_rows_and_zero_tables_are_exact():
    rng = np.random.RandomState(1)
    P, Q = rng.randn(8, 16).astype(np.float32), rng.randn(40, 16).astype(np.float32)
    P[3] = 0.0
    M, eps, V = _filter(P, Q)
    # (the scales are rounded UP for the bound: a zero scale becomes the smallest sub-normal, nothing more)
    assert (V[3] == 0).all() and eps[3] < 1e-40 and (np.abs(M[3]) < 1e-35).all()
    M0, eps0, V0 = _filter(P, np.zeros_like(Q))
    assert (V0 == 0).all() and (np.abs(M0) < 1e-35).all()
