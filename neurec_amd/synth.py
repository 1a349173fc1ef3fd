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
