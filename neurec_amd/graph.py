"""Host-side construction of the normalised bipartite adjacency (one-off, at model init).

Mirrors LightGCN.create_adj_mat (model/general_recommender/LightGCN.py:34-78) and the
default `norm` branch of NGCF.get_adj_mat (NGCF.py:299-318): the result is the CSR matrix the
SpMM kernel consumes — N = n_users + n_items nodes, user rows first, ascending columns,
fp32 values — with the reference's rounding: every value is produced by the same sequence
of fp32 (or, where the reference silently promotes through `sp.eye`, fp64) operations.

Pinned: tests/test_adjacency_golden.py compares every adj_type bit-for-bit with matrices made by
the reference's own methods (tests/golden/make_golden_adjacency.py).  One order difference is
known: for LightGCN `gcmc` the reference's single scipy product leaves each row's columns in
DESCENDING order and hands TF the COO that way.  By default every adjacency here is canonical
(ascending columns) — same entries, same values; `lightgcn_adjacency(..., tf_order=True)` returns the
`gcmc` rows in the reference's descending storage order (pass `keep_order=True` on to
LightGCNEngine / SpmmCSR.from_scipy: the SpMM kernels sum a row in its storage order), so that mode
can be bit-faithful too.

This runs once per training run on the host, exactly where the reference runs it; it is not
on the measured path.
"""
import numpy as np
import scipy.sparse as sp

ADJ_TYPES = ("plain", "norm", "gcmc", "pre", "mean")


def _inv_power(rowsum, p):
    with np.errstate(divide="ignore"):
        out = np.power(rowsum, p)
    out[np.isinf(out)] = 0.0                      # isolated nodes: inf -> 0 (LightGCN.py:66-67)
    return out


def bipartite_adjacency(user_idx, item_idx, n_users, n_items):
    """A = R (+) R^T on N nodes as a canonical fp32 CSR with unit entries (LightGCN.py:36-42)."""
    u = np.asarray(user_idx, dtype=np.int64)
    i = np.asarray(item_idx, dtype=np.int64) + n_users
    n = n_users + n_items
    rows = np.concatenate([u, i])
    cols = np.concatenate([i, u])
    a = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return a


def lightgcn_adjacency(user_idx, item_idx, n_users, n_items, adj_type="pre", tf_order=False):
    """tf_order=True: rows stored in the order the reference hands them to TensorFlow (differs from
    ascending columns for `gcmc` only: descending)."""
    a = bipartite_adjacency(user_idx, item_idx, n_users, n_items)
    n = a.shape[0]
    row_of = np.repeat(np.arange(n), np.diff(a.indptr))
    if adj_type == "plain":
        out = a
    elif adj_type == "pre":
        # D^-1/2 A D^-1/2, all fp32: (d_r^-1/2 * a_rc) * d_c^-1/2
        deg = np.asarray(a.sum(1)).ravel().astype(np.float32)
        dinv = _inv_power(deg, np.float32(-0.5)).astype(np.float32)
        vals = (dinv[row_of] * a.data) * dinv[a.indices]
        out = sp.csr_matrix((vals.astype(np.float32), a.indices.copy(), a.indptr.copy()), shape=a.shape)
    elif adj_type == "gcmc":
        deg = np.asarray(a.sum(1)).ravel().astype(np.float32)
        dinv = _inv_power(deg, np.float32(-1)).astype(np.float32)
        out = sp.csr_matrix(((dinv[row_of] * a.data).astype(np.float32), a.indices.copy(),
                             a.indptr.copy()), shape=a.shape)
    elif adj_type == "norm":
        # D^-1 (A + I): `sp.eye` is float64, so the reference computes this branch in fp64
        # and only rounds to fp32 when it builds the sparse tensor (LightGCN.py:152).
        ai = (a.astype(np.float64) + sp.eye(n, dtype=np.float64)).tocsr()
        ai.sort_indices()
        deg = np.asarray(ai.sum(1)).ravel()
        dinv = _inv_power(deg, -1.0)
        r_of = np.repeat(np.arange(n), np.diff(ai.indptr))
        out = sp.csr_matrix(((dinv[r_of] * ai.data).astype(np.float32), ai.indices, ai.indptr),
                            shape=ai.shape)
    elif adj_type == "mean":
        deg = np.asarray(a.sum(1)).ravel().astype(np.float32)
        dinv = _inv_power(deg, np.float32(-1)).astype(np.float32)
        mean_adj = sp.csr_matrix(((dinv[row_of] * a.data).astype(np.float32), a.indices.copy(),
                                  a.indptr.copy()), shape=a.shape)
        out = (mean_adj.astype(np.float64) + sp.eye(n, dtype=np.float64)).tocsr().astype(np.float32)
    else:
        raise ValueError("adj_type must be one of %s" % (ADJ_TYPES,))
    out = out.tocsr().astype(np.float32)
    out.sort_indices()
    if tf_order and adj_type == "gcmc":
        out = reverse_rows(out)
    return out


def reverse_rows(a):
    """the same CSR with every row's entries in reverse storage order (ascending -> descending columns)"""
    a = a.tocsr()
    nnz = a.nnz
    row_of = np.repeat(np.arange(a.shape[0]), np.diff(a.indptr))
    # position p of row r (p in [lo, hi)) takes the entry at lo + hi - 1 - p
    src = a.indptr[row_of] + a.indptr[row_of + 1] - 1 - np.arange(nnz)
    out = sp.csr_matrix((a.data[src], a.indices[src], a.indptr.copy()), shape=a.shape)
    out.has_sorted_indices = False
    return out


def transpose_csr(a):
    t = a.T.tocsr().astype(np.float32)
    t.sort_indices()
    return t


def is_symmetric(a):
    return (a != a.T).nnz == 0


def ngcf_adjacency(train_matrix, adj_type="norm"):
    """NGCF.get_adj_mat (NGCF.py:299-318).  Unlike LightGCN, the bipartite block carries the
    *values* of the train matrix (`self.graph = dataset.train_matrix.toarray()`, NGCF.py:40), and
    the default `norm` type is the row-normalised D^-1 (A + I) — not symmetric, so the backward
    pass needs the transpose.  Built sparsely (the reference densifies U×I, ≈9.8 GB at gowalla)."""
    r = sp.csr_matrix(train_matrix, dtype=np.float32)
    r.sum_duplicates()
    n = r.shape[0] + r.shape[1]
    a = sp.bmat([[None, r], [r.T.tocsr(), None]], format="csr", dtype=np.float32)
    a.sort_indices()

    def row_normalised(m):                         # d_inv follows m's dtype (fp64 once eye is added)
        deg = np.asarray(m.sum(1)).ravel()
        dinv = _inv_power(deg, -1)
        rows = np.repeat(np.arange(m.shape[0]), np.diff(m.indptr))
        return sp.csr_matrix((dinv[rows] * m.data, m.indices.copy(), m.indptr.copy()), shape=m.shape)

    if adj_type == "plain":
        out = a
    elif adj_type == "norm":
        ai = (a.astype(np.float64) + sp.eye(n, dtype=np.float64)).tocsr()
        ai.sort_indices()
        out = row_normalised(ai)
    elif adj_type == "gcmc":
        out = row_normalised(a)
    else:
        out = (row_normalised(a).astype(np.float64) + sp.eye(n, dtype=np.float64)).tocsr()
    out = out.tocsr().astype(np.float32)
    out.sort_indices()
    return out
