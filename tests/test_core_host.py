"""neurec_amd/csrc/nr_core.h — the per-thread arithmetic the HIP kernels inline —
compiled for the host (tests/hostcheck) and checked against oracle/ on the CPU."""
import numpy as np
import pytest

from conftest import golden_eval_cases, load_golden, truth_lists
from oracle import native, train


def test_heap_emulation_equals_libstdcxx_partial_sort_copy(hostcheck):
    rng = np.random.RandomState(11)
    for t in range(400):
        n = int(rng.randint(1, 500)); k = int(rng.randint(1, min(n, 128) + 1))
        s = rng.randn(n).astype(np.float32)
        mode = t % 5
        if mode == 1: s = (np.round(s * 2) / 2).astype(np.float32)
        if mode == 2: s[:] = 0; s[rng.randint(0, n, size=n // 3)] = 1
        if mode == 3: s[rng.rand(n) < 0.5] = -np.inf
        if mode == 4: s[::2] = 0.0; s[1::4] = -0.0
        out = np.zeros(k, np.int32)
        hostcheck.hc_partial_sort_copy(s, n, k, out)
        np.testing.assert_array_equal(out, native.arg_topk(s[None, :], k, threads=1)[0])


@pytest.mark.parametrize("case", golden_eval_cases())
def test_metric_formulas_bit_exact_on_reference_golden(hostcheck, case):
    g = load_golden(case)
    K = int(g["top_k"]); truth = truth_lists(g["truth_ptr"], g["truth_idx"])
    metrics = g["metrics"].tolist()
    for r in range(g["scores"].shape[0]):
        rank = np.zeros(min(2 * K, g["scores"].shape[1]), np.int32)
        hostcheck.hc_partial_sort_copy(np.ascontiguousarray(g["scores"][r]), g["scores"].shape[1],
                                       len(rank), rank)
        hits = np.isin(rank[:K], truth[r]).astype(np.uint8)
        for mi, mid in enumerate(metrics):
            out = np.zeros(K, np.float32)
            hostcheck.hc_metric(mid, hits, K, len(truth[r]), out)
            np.testing.assert_array_equal(out, g["result"][r, mi * K:(mi + 1) * K])


def test_key_packing_orders_like_the_reference_comparator(hostcheck):
    vals = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-30, 2.0, np.inf], np.float32)
    keys = [hostcheck.hc_pack_key(float(v), 5) for v in vals]
    assert keys[3] == keys[4]                              # -0.0 ties with +0.0
    ks = [k for i, k in enumerate(keys) if i != 3]
    assert ks == sorted(ks)
    assert hostcheck.hc_pack_key(1.0, 3) > hostcheck.hc_pack_key(1.0, 4)    # lower index first
    for v in vals:
        k = hostcheck.hc_pack_key(float(v), 123456)
        assert hostcheck.hc_key_index(k) == 123456
        assert hostcheck.hc_key_score(k) == (v + np.float32(0))


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 1000, 4097, 80367])
def test_epoch_permutation_is_a_bijection(hostcheck, n):
    for key in (1, 0xdeadbeef):
        perm = np.array([hostcheck.hc_permute_index(i, n, key) for i in range(n)], np.int64)
        assert perm.min() == 0 and perm.max() == n - 1
        assert len(np.unique(perm)) == n
    if n >= 1000:
        a = np.array([hostcheck.hc_permute_index(i, n, 1) for i in range(n)])
        b = np.array([hostcheck.hc_permute_index(i, n, 2) for i in range(n)])
        assert (a != b).mean() > 0.99                      # different epoch keys, different order
        assert (a != np.arange(n)).mean() > 0.99
        # displacement looks uniform: mean |perm(i)-i| ~ n/3
        assert abs(np.abs(a - np.arange(n)).mean() / n - 1 / 3) < 0.03


def test_negative_draws_respect_exclusion_and_are_uniform(hostcheck):
    rng = np.random.RandomState(5)
    high = 50
    excl = np.sort(rng.choice(high, 20, replace=False)).astype(np.int32)
    draws = np.array([hostcheck.hc_draw_negative(2018, 3, c, high, excl, len(excl))
                      for c in range(30000)])
    assert not np.isin(draws, excl).any() and draws.min() >= 0 and draws.max() < high
    counts = np.bincount(draws, minlength=high)[np.setdiff1d(np.arange(high), excl)]
    chi2 = ((counts - counts.mean()) ** 2 / counts.mean()).sum()
    assert chi2 < 80                                        # 29 dof, p ~ 1e-6
    # nearly-full user: the bounded fallback still returns the only allowed ids
    full = np.setdiff1d(np.arange(high), [7, 31]).astype(np.int32)
    got = {hostcheck.hc_draw_negative(1, 0, c, high, full, len(full)) for c in range(200)}
    assert got == {7, 31}
    assert hostcheck.hc_draw_negative(1, 0, 0, high, np.arange(high, dtype=np.int32), high) == -1
    allowed = np.setdiff1d(np.arange(high), excl)
    for r in range(len(allowed)):
        assert hostcheck.hc_nth_allowed(r, excl, len(excl)) == allowed[r]


def test_adam_and_bpr_scalars_match_oracle(hostcheck):
    rng = np.random.RandomState(9)
    n = 4096
    g = (rng.randn(n) * 0.1).astype(np.float32)
    for sparse in (False, True):
        var = rng.randn(n).astype(np.float32); m = (rng.randn(n) * 0.01).astype(np.float32)
        v = (rng.rand(n) * 0.01).astype(np.float32)
        ov, om, ovv = var.copy(), m.copy(), v.copy()
        ad = train.Adam(0.01)
        for _ in range(3):
            ad.advance()
        gg = g.copy()
        if sparse:
            gg[::3] = 0
            hostcheck.hc_adam_sparse(n, gg, var, m, v, float(ad.alpha()), 0.9, 0.999, 1e-8)
            ad.sparse_swept(ov, om, ovv, gg)
        else:
            hostcheck.hc_adam_dense(n, gg, var, m, v, float(ad.alpha()), 0.9, 0.999, 1e-8)
            ad.dense(ov, om, ovv, gg)
        np.testing.assert_array_equal(m, om)               # same roundings, bit for bit
        np.testing.assert_array_equal(v, ovv)
        np.testing.assert_array_equal(var, ov)
    x = np.linspace(-30, 30, 601).astype(np.float32)
    lb, gr = train.bpr_terms(x)
    got_l = np.array([hostcheck.hc_bpr_loss(float(t)) for t in x], np.float32)
    got_g = np.array([hostcheck.hc_bpr_dloss(float(t)) for t in x], np.float32)
    np.testing.assert_allclose(got_l, lb, rtol=3e-7, atol=1e-30)
    np.testing.assert_allclose(got_g, gr, rtol=3e-7, atol=1e-30)
