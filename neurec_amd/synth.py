"""Synthetic interaction matrices with the shapes of the BASELINE configs.

`gowalla.train` is absent from the reference tree (.MISSING_LARGE_BLOBS) and there is no
network, so throughput runs use a synthetic twin with gowalla's size (U=29,858, I=40,981,
E~810k), a log-normal user-degree law and power-law item popularity.  Host-side numpy only:
this is input generation, not part of the measured path.
"""
import numpy as np
import scipy.sparse as sp

SHAPES = {
    # name: (n_users, n_items, n_train, mean_test_per_user)
    "ml-100k": (943, 1682, 80367, 20.8),
    "gowalla": (29858, 40981, 810128, 7.28),
}


def _draw_unique(rng, n_items, degrees, cdf, forbid=None):
    """Per user, `degrees[u]` distinct items drawn from the popularity law `cdf`."""
    U = len(degrees)
    over = (degrees * 1.6).astype(np.int64) + 8
    owner = np.repeat(np.arange(U, dtype=np.int64), over)
    items = np.searchsorted(cdf, rng.random_sample(owner.shape[0])).astype(np.int64)
    items = np.minimum(items, n_items - 1)
    key = owner * n_items + items
    if forbid is not None:
        key = key[~np.isin(key, forbid)]
    key = np.unique(key)                                  # dedup, sorted by (user, item)
    owner, items = key // n_items, key % n_items
    # keep at most degrees[u] per user, chosen at random among the distinct draws
    order = np.lexsort((rng.random_sample(len(owner)), owner))
    owner, items = owner[order], items[order]
    start = np.searchsorted(owner, np.arange(U))
    rank = np.arange(len(owner)) - start[owner]
    keep = rank < degrees[owner]
    return owner[keep].astype(np.int32), items[keep].astype(np.int32)


def interactions_around_test(test, n_train, seed=2018):
    """SURVEY §8d's gowalla-shaped workload: the test split is GIVEN (the reference's real
    dataset/gowalla.test, 217,242 pairs — tests/golden/gowalla_test_split.npz) and the missing train side
    is synthesised around it: user degrees max(8, round(LogNormal(2.9, 0.9))) rescaled to n_train
    interactions, items drawn without replacement per user with p ∝ (rank + 10)^-0.8, never one of the user's
    test items.  An item's rank is its POPULARITY RANK IN THE GIVEN TEST SPLIT (ties in shuffled order): as in a
    real dataset the train and test popularities go together, so a trained model has something to find and the
    run's NDCG@10 is a number about ranking, not about noise (r04: ranks were shuffled — independent of the test
    split — and NDCG@10 stayed at chance).  The multiset of popularities, hence the degree law and the hub
    rows, is the same as before.  Returns (train_csr, test_csr)."""
    test = sp.csr_matrix(test, dtype=np.float32)
    test.sort_indices()
    U, I = test.shape
    rng = np.random.RandomState(seed)
    deg = np.maximum(8, np.round(rng.lognormal(2.9, 0.9, U))).astype(np.float64)
    deg = np.maximum(1, np.round(deg * (n_train / deg.sum()))).astype(np.int64)
    deg = np.minimum(deg, I // 4)
    coo = test.tocoo()
    shuffled = rng.permutation(I)
    by_test_count = shuffled[np.argsort(-np.bincount(coo.col, minlength=I)[shuffled], kind="stable")]
    pop = np.empty(I)
    pop[by_test_count] = (np.arange(I) + 10.0) ** -0.8           # the most tested item is the most popular
    cdf = np.cumsum(pop / pop.sum())
    tu, ti = _draw_unique(rng, I, deg, cdf, forbid=np.sort(coo.row.astype(np.int64) * I + coo.col))
    train = sp.csr_matrix((np.ones(len(tu), np.float32), (tu, ti)), shape=(U, I))
    train.sort_indices()
    return train, test


def load_test_split(path):
    """the CSR fixture written by tests/golden/make_gowalla_test_fixture.py"""
    z = np.load(path)
    U, I = (int(x) for x in z["shape"])
    return sp.csr_matrix((np.ones(len(z["indices"]), np.float32), z["indices"], z["indptr"]), shape=(U, I))


def interactions(shape="gowalla", seed=2018, scale=1.0):
    """Returns (train_csr, test_csr) scipy matrices; `scale` multiplies users/items/edges."""
    U, I, E, mean_test = SHAPES[shape]
    U, I, E = int(U * scale), int(I * scale), int(E * scale)
    rng = np.random.RandomState(seed)
    deg = np.maximum(8, np.round(rng.lognormal(2.9, 0.9, U))).astype(np.float64)
    deg = np.maximum(1, np.round(deg * (E / deg.sum()))).astype(np.int64)
    deg = np.minimum(deg, I // 4)
    pop = (np.arange(I) + 10.0) ** -0.8
    rng.shuffle(pop)                                      # popularity is not tied to the item id
    cdf = np.cumsum(pop / pop.sum())
    tu, ti = _draw_unique(rng, I, deg, cdf)
    train = sp.csr_matrix((np.ones(len(tu), np.float32), (tu, ti)), shape=(U, I))
    n_test = np.clip(rng.geometric(1.0 / mean_test, U), 1, 200).astype(np.int64)
    su, si = _draw_unique(rng, I, n_test, cdf, forbid=tu.astype(np.int64) * I + ti)
    test = sp.csr_matrix((np.ones(len(su), np.float32), (su, si)), shape=(U, I))
    train.sort_indices()
    test.sort_indices()
    return train, test


def xavier_uniform(rows, d, rng):
    """tf.contrib.layers.xavier_initializer() for a [rows, d] variable (LightGCN.py:87-89) [EXT]."""
    lim = np.sqrt(6.0 / (rows + d))
    return rng.uniform(-lim, lim, (rows, d)).astype(np.float32)


# ----------------------------------------------------------------------------- on the device
CONFIG4 = (10_000_000, 1_000_000, 200_000_000)      # BASELINE configs[3]: users, items, train edges


def device_interactions(n_users, n_items, n_edges, seed=2018, device="cuda", max_degree=2000):
    """The interaction law of `interactions` (log-normal user degrees floored at 8 and rescaled to
    n_edges, capped; items by inverse-CDF draws from p ∝ (rank + 10)^-0.8 with shuffled ranks),
    generated ON THE DEVICE from torch's counter-based Philox stream — SURVEY §8d asks that the
    config-4 graph (2·10⁸ edges) is never materialised on the host.  Duplicate (user, item) draws
    are dropped, so the edge count comes out a little under n_edges.  Returns device tensors
    (indptr int64 [U+1], item indices int32 ascending per user)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    deg = torch.exp(torch.randn(n_users, generator=g, device=device) * 0.9 + 2.9).round().clamp_(min=8)
    deg = (deg * (n_edges / float(deg.sum()))).round().clamp_(1, min(max_degree, n_items // 4)).long()
    pop = (torch.arange(n_items, device=device, dtype=torch.float64) + 10.0) ** -0.8
    pop = pop[torch.randperm(n_items, generator=g, device=device)]           # popularity is not tied to the id
    cdf = torch.cumsum(pop / pop.sum(), 0).float()
    owner = torch.repeat_interleave(torch.arange(n_users, device=device), deg)
    items = torch.searchsorted(cdf, torch.rand(owner.numel(), generator=g, device=device)).clamp_(max=n_items - 1)
    key = torch.unique(owner * n_items + items)                               # dedup; sorted by (user, item)
    del owner, items
    users = torch.div(key, n_items, rounding_mode="floor")
    indices = (key - users * n_items).to(torch.int32)
    indptr = torch.zeros(n_users + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(torch.bincount(users, minlength=n_users), 0)
    return indptr, indices


def device_lightgcn_adjacency(indptr, indices, n_users, n_items, row_lo=0, row_hi=None):
    """Rows [row_lo, row_hi) of the `pre` adjacency D^-1/2 A D^-1/2 (LightGCN.py:63-72, the fp32
    rounding of graph.lightgcn_adjacency: (d_r^-1/2 · 1) · d_c^-1/2) of the bipartite graph given as a
    device CSR of the U x I train matrix; N = U + I nodes, user rows first.  Device tensors
    (indptr int64 relative to the block, indices int32 global columns ascending, vals fp32) — each
    rank of a row-sharded run builds only its own block."""
    import torch
    dev = indptr.device
    N = n_users + n_items
    row_hi = N if row_hi is None else row_hi
    nnz_u = int(indptr[-1])
    deg_u = (indptr[1:] - indptr[:-1])
    deg_i = torch.bincount(indices[:nnz_u].long(), minlength=n_items)
    # d^-1/2 exactly as graph.lightgcn_adjacency makes it (numpy's fp32 power — torch's differs in the
    # last ulp on a third of the degrees): the N degrees take the host detour, the edges never do
    from .graph import _inv_power
    deg = torch.cat([deg_u, deg_i]).cpu().numpy().astype(np.float32)
    dinv = torch.from_numpy(_inv_power(deg, np.float32(-0.5)).astype(np.float32)).to(dev)
    parts_ptr, parts_idx, parts_val = [], [], []
    # user rows of the block: columns U + item, already ascending
    ulo, uhi = min(row_lo, n_users), min(row_hi, n_users)
    if uhi > ulo:
        b, e = int(indptr[ulo]), int(indptr[uhi])
        cols = indices[b:e].long() + n_users
        rows = torch.repeat_interleave(torch.arange(ulo, uhi, device=dev), deg_u[ulo:uhi], output_size=e - b)
        parts_idx.append(cols.to(torch.int32))
        parts_val.append((dinv[rows] * 1.0) * dinv[cols])
        parts_ptr.append(deg_u[ulo:uhi])
    # item rows of the block: the transpose, users ascending per item
    ilo, ihi = max(row_lo, n_users) - n_users, max(row_hi, n_users) - n_users
    if ihi > ilo:
        it = indices[:nnz_u].long()
        sel = (it >= ilo) & (it < ihi)
        us = torch.repeat_interleave(torch.arange(n_users, device=dev), deg_u, output_size=nnz_u)[sel]
        it = it[sel]
        order = torch.argsort(it * n_users + us)                             # by (item, user)
        it, us = it[order], us[order]
        parts_idx.append(us.to(torch.int32))
        parts_val.append((dinv[it + n_users] * 1.0) * dinv[us])
        parts_ptr.append(deg_i[ilo:ihi])
    counts = torch.cat(parts_ptr) if parts_ptr else torch.zeros(0, dtype=torch.int64, device=dev)
    ptr = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(counts, 0)
    idx = torch.cat(parts_idx) if parts_idx else torch.zeros(0, dtype=torch.int32, device=dev)
    val = torch.cat(parts_val) if parts_val else torch.zeros(0, dtype=torch.float32, device=dev)
    return ptr, idx, val


def device_lightgcn_rank_rows(indptr, indices, n_users, n_items, user_range, item_range):
    """CSR of a rank's rows under parallel.BipartitePartition: its user rows, then its item rows
    (global node ids as columns) — the `local_rows=` input of sharded.ShardedLightGCN."""
    import torch
    pu, iu, vu = device_lightgcn_adjacency(indptr, indices, n_users, n_items, user_range[0], user_range[1])
    pi, ii, vi = device_lightgcn_adjacency(indptr, indices, n_users, n_items, n_users + item_range[0],
                                           n_users + item_range[1])
    return torch.cat([pu, pi[1:] + pu[-1]]), torch.cat([iu, ii]), torch.cat([vu, vi])


def device_test_rows(train_rows, n_items, per_user=2, seed=2019):
    """A synthetic test split for a block of users of a device-generated graph (the config-4 evaluation leg):
    `per_user` uniform item draws per user, duplicates and the user's TRAIN items dropped (dataset.py's splits are
    disjoint) — so a few users end up with fewer, or no, test items and are skipped like tool.py:63 skips them.
    train_rows: engine.DeviceCSR of those users.  Returns an engine.DeviceCSR with the same rows."""
    import torch
    from . import engine as E
    dev = train_rows.indptr.device
    n = train_rows.n_rows
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    owner = torch.arange(n, device=dev).repeat_interleave(per_user)
    items = torch.randint(0, n_items, (owner.numel(),), generator=g, device=dev)
    key = torch.unique(owner * n_items + items)
    if train_rows.nnz:
        tkey = train_rows.row_of().long() * n_items + train_rows.indices[:train_rows.nnz].long()   # ascending
        pos = torch.searchsorted(tkey, key).clamp_(max=tkey.numel() - 1)
        key = key[tkey[pos] != key]
    users = torch.div(key, n_items, rounding_mode="floor")
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(torch.bincount(users, minlength=n), 0)
    idx = (key - users * n_items).to(torch.int32)
    if idx.numel() == 0:
        idx = torch.zeros(1, dtype=torch.int32, device=dev)
    return E.DeviceCSR(indptr, idx, n_items)
