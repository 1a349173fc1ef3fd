"""HIP BPR sampler: structure of the epoch stream (the reference's invariants,
data/sampler.py:24-39,71-90,198-206), bit-exactness against the host build of the same
per-thread code, and distribution of the negatives."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from neurec_amd import engine
    return engine


def _toy(eng, rng, U=300, I=500, lo=1, hi=40):
    deg = rng.randint(lo, hi, U)
    deg[rng.rand(U) < 0.1] = 0                       # users without train items never appear
    indptr = np.zeros(U + 1, np.int64); indptr[1:] = np.cumsum(deg)
    indices = np.concatenate([np.sort(rng.choice(I, n, replace=False)) for n in deg] +
                             [np.zeros(0, np.int64)]).astype(np.int32)
    return eng.DeviceCSR(indptr, indices, I), indptr, indices


def test_epoch_stream_structure(eng):
    rng = np.random.RandomState(0)
    csr, indptr, indices = _toy(eng, rng)
    E, I = csr.nnz, 500
    row_of = csr.row_of()
    u, p, n = [t.cpu().numpy() for t in eng.sample_bpr_epoch(csr, row_of, I, 1, 2018, 0, True)]
    assert len(u) == len(p) == len(n) == E
    # every positive exactly once per epoch (one permutation of E, data_iterator.py:58-60)
    pairs = np.sort(u.astype(np.int64) * I + p)
    want = np.sort(np.repeat(np.arange(len(indptr) - 1), np.diff(indptr)).astype(np.int64) * I + indices)
    np.testing.assert_array_equal(pairs, want)
    # negatives are valid ids the user has not interacted with (sampler.py:80-81)
    assert n.min() >= 0 and n.max() < I
    for q in range(0, E, 37):
        assert n[q] not in indices[indptr[u[q]]:indptr[u[q] + 1]]
    # shuffle=False: user-major, item-ascending order of users_list / pos_items_list
    u0, p0, _ = [t.cpu().numpy() for t in eng.sample_bpr_epoch(csr, row_of, I, 1, 2018, 0, False)]
    np.testing.assert_array_equal(u0, np.repeat(np.arange(len(indptr) - 1), np.diff(indptr)))
    np.testing.assert_array_equal(p0, indices)
    # same (seed, epoch) -> same stream; next epoch -> different order and negatives
    u1, p1, n1 = [t.cpu().numpy() for t in eng.sample_bpr_epoch(csr, row_of, I, 1, 2018, 0, True)]
    np.testing.assert_array_equal(u1, u); np.testing.assert_array_equal(n1, n)
    u2, p2, n2 = [t.cpu().numpy() for t in eng.sample_bpr_epoch(csr, row_of, I, 1, 2018, 1, True)]
    assert (u2 != u).mean() > 0.9 and not np.array_equal(n2, n)
    # a slice of the stream equals the same slice of the whole (multi-GPU sharding relies on it)
    us, ps, ns = [t.cpu().numpy() for t in eng.sample_bpr_epoch(csr, row_of, I, 1, 2018, 0, True,
                                                                 begin=1000, count=777)]
    np.testing.assert_array_equal(us, u[1000:1777]); np.testing.assert_array_equal(ns, n[1000:1777])


def test_device_stream_equals_host_build_of_the_same_code(eng, hostcheck):
    rng = np.random.RandomState(1)
    csr, indptr, indices = _toy(eng, rng, U=60, I=90, hi=30)
    E, I, neg_num, seed, epoch = csr.nnz, 90, 3, 77, 5
    u, p, n = [t.cpu().numpy() for t in eng.sample_bpr_epoch(csr, csr.row_of(), I, neg_num, seed,
                                                             epoch, True)]
    row_of = np.repeat(np.arange(60), np.diff(indptr))
    # splitmix64 in python to derive the permutation key exactly as the kernel does
    M = (1 << 64) - 1

    def sm(x):
        x = (x + 0x9e3779b97f4a7c15) & M
        x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & M
        x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & M
        return x ^ (x >> 31)
    key = sm(seed ^ sm((epoch + 0x51ed27) & M))
    for q in range(0, E, 11):
        t = hostcheck.hc_permute_index(q, E, key)
        assert u[q] == row_of[t] and p[q] == indices[t]
        ex = np.ascontiguousarray(indices[indptr[u[q]]:indptr[u[q] + 1]])
        for k in range(neg_num):
            assert n[q * neg_num + k] == hostcheck.hc_draw_negative(seed, epoch, t * neg_num + k, I,
                                                                    ex, len(ex))


def test_negatives_uniform_over_non_interacted_items(eng):
    rng = np.random.RandomState(2)
    U, I = 4, 64
    lists = [np.sort(rng.choice(I, 20, replace=False)) for _ in range(U)]
    indptr = np.arange(0, 20 * U + 1, 20, dtype=np.int64)
    csr = eng.DeviceCSR(indptr, np.concatenate(lists).astype(np.int32), I)
    row_of = csr.row_of()
    counts = np.zeros((U, I))
    for ep in range(400):
        u, p, n = [t.cpu().numpy() for t in eng.sample_bpr_epoch(csr, row_of, I, 4, 9, ep, True)]
        np.add.at(counts, (np.repeat(u, 4), n), 1)
    for uu in range(U):
        assert counts[uu, lists[uu]].sum() == 0
        c = counts[uu, np.setdiff1d(np.arange(I), lists[uu])]
        chi2 = ((c - c.mean()) ** 2 / c.mean()).sum()
        assert chi2 < 100, chi2                                   # 43 dof


def test_batch_randint_choice_kernel(eng):
    rng = np.random.RandomState(3)
    high = 200
    excl = [np.sort(rng.choice(high, rng.randint(0, 150), replace=False)).tolist() for _ in range(50)]
    sizes = [int(rng.randint(1, 40)) for _ in range(50)]
    from oracle.native import lists_to_csr
    ptr, idx = lists_to_csr(excl)
    ecsr = eng.DeviceCSR(ptr, idx[:int(ptr[-1])], high)
    for replace in (True, False):
        out, off = eng.randint_choice_batch(high, sizes, ecsr, replace, 5, 1)
        out = out.cpu().numpy()
        assert len(out) == sum(sizes)
        for q in range(50):
            got = out[off[q]:off[q + 1]]
            assert got.min() >= 0 and got.max() < high and not np.isin(got, excl[q]).any()
            if not replace:
                assert len(set(got.tolist())) == len(got)
    o1, _ = eng.randint_choice_batch(high, sizes, ecsr, True, 5, 1)
    o2, _ = eng.randint_choice_batch(high, sizes, ecsr, True, 5, 2)
    assert not np.array_equal(o1.cpu().numpy(), o2.cpu().numpy())


def test_sampler_front_end_batches_and_errors(eng):
    from neurec_amd.trainer import BprEpochSampler
    rng = np.random.RandomState(4)
    csr, indptr, indices = _toy(eng, rng, U=100, I=300)
    s = BprEpochSampler(csr, 300, neg_num=1, batch_size=256, shuffle=True, seed=1)
    sizes = [int(u.numel()) for u, p, n in s.batches()]
    assert len(sizes) == len(s) == (csr.nnz + 255) // 256
    assert sizes[:-1] == [256] * (len(sizes) - 1) and sum(sizes) == csr.nnz   # last short batch kept
    s3 = BprEpochSampler(csr, 300, neg_num=3, batch_size=100)
    u, p, n = next(iter(s3.batches()))
    assert tuple(n.shape) == (100, 3)
    with pytest.raises(ValueError):
        BprEpochSampler(csr, 300, neg_num=0)
    with pytest.raises(ValueError):
        BprEpochSampler(csr, 10)                 # a user has >= n_items interactions
    # rank slices partition the epoch
    parts = [BprEpochSampler(csr, 300, batch_size=64, seed=3, rank=r, world=3) for r in range(3)]
    whole = BprEpochSampler(csr, 300, batch_size=64, seed=3)
    wu = whole.sample_epoch()[0].cpu().numpy()
    cat = np.concatenate([p_.sample_epoch()[0].cpu().numpy() for p_ in parts])
    np.testing.assert_array_equal(cat, wu)
