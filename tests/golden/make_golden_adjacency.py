"""Golden adjacency matrices produced by the REFERENCE's own builders:

  LightGCN.create_adj_mat        model/general_recommender/LightGCN.py:34-78   (all five adj_type)
  LightGCN._convert_sp_mat_to_sp_tensor  :151-154  (the COO + fp32 cast handed to TF)
  NGCF.get_adj_mat / normalized_adj_single   model/general_recommender/NGCF.py:289-318

The two model files import TensorFlow at module level, so the methods are lifted out of the
reference source with `ast` (decorators dropped) and executed as they are, bound to a stub `self`
that carries only what they read (dataset.get_train_interactions / n_users / n_items; graph /
num_users / num_items / adj_type / logger).  `tf.SparseTensor` is replaced by a tuple so that the
conversion routine's own `tocoo().astype(np.float32)` is what produces the stored values.

Run in the build container (needs /root/reference):

    python tests/golden/make_golden_adjacency.py

Writes tests/golden/adj_{plain,norm,gcmc,pre,mean}.npz and ngcf_adj_{plain,norm,gcmc,mean}.npz:
inputs (users, items, n_users, n_items [, values]) + the reference COO (row, col, data) in the
order the reference hands it to TensorFlow.
"""
import ast
import os
import types

import numpy as np
import scipy.sparse as sp

REF = "/root/reference/model/general_recommender"
OUT = os.path.dirname(os.path.abspath(__file__))


class _NumpyWithMat:
    """numpy as the reference's pinned 1.16 saw it: `np.mat` (LightGCN.py:153) left numpy in 2.0."""
    mat = staticmethod(np.asmatrix)

    def __getattr__(self, name):
        return getattr(np, name)


def reference_methods(path, cls, names, extra_ns):
    """{name: function} for the methods `names` of class `cls` in the reference file `path`."""
    tree = ast.parse(open(path).read())
    ns = {"np": _NumpyWithMat(), "sp": sp}
    ns.update(extra_ns)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name in names:
                    item.decorator_list = []                    # @timer comes from util (imports TF)
                    mod = ast.Module([item], [])
                    ast.fix_missing_locations(mod)
                    exec(compile(mod, path, "exec"), ns)
    return {n: ns[n] for n in names}


def toy_interactions(n_users, n_items, seed):
    """A small power-law bipartite graph with isolated users and items (degree 0 -> the
    inf -> 0 branch of LightGCN.py:66-67) and one hub item."""
    rs = np.random.RandomState(seed)
    users, items = [], []
    for u in range(n_users):
        if u % 17 == 5:
            continue                                            # isolated user
        deg = int(min(n_items - 3, max(1, round(rs.lognormal(1.6, 0.9)))))
        p = 1.0 / (np.arange(n_items - 2) + 4.0) ** 0.8        # the last two items stay isolated
        its = rs.choice(n_items - 2, deg, replace=False, p=p / p.sum())
        users += [u] * deg
        items += sorted(int(i) for i in its)
    return np.asarray(users, np.int32), np.asarray(items, np.int32)


def main():
    n_users, n_items = 157, 131
    users, items = toy_interactions(n_users, n_items, 2018)

    # ---- LightGCN
    tf_stub = types.SimpleNamespace(SparseTensor=lambda idx, data, shape: (np.asarray(idx), data, shape))
    m = reference_methods(os.path.join(REF, "LightGCN.py"), "LightGCN",
                          ["create_adj_mat", "_convert_sp_mat_to_sp_tensor"], {"tf": tf_stub})
    me = types.SimpleNamespace(
        n_users=n_users, n_items=n_items,
        dataset=types.SimpleNamespace(get_train_interactions=lambda: (users.tolist(), items.tolist())))
    for adj_type in ("plain", "norm", "gcmc", "pre", "mean"):
        with np.errstate(divide="ignore"):
            adj = m["create_adj_mat"](me, adj_type)
        idx, data, shape = m["_convert_sp_mat_to_sp_tensor"](me, adj)
        assert data.dtype == np.float32 and tuple(shape) == (n_users + n_items,) * 2
        np.savez_compressed(os.path.join(OUT, "adj_%s.npz" % adj_type), users=users, items=items,
                            n_users=n_users, n_items=n_items, row=idx[:, 0].astype(np.int64),
                            col=idx[:, 1].astype(np.int64), data=data)
        print("adj_%s: nnz=%d" % (adj_type, len(data)))

    # ---- NGCF (bipartite block carries the train matrix's values: NGCF.py:40,302-303)
    m = reference_methods(os.path.join(REF, "NGCF.py"), "NGCF",
                          ["get_adj_mat", "normalized_adj_single", "_convert_sp_mat_to_sp_tensor"],
                          {"tf": tf_stub})
    rs = np.random.RandomState(7)
    values = rs.choice(np.asarray([1.0, 2.0, 3.0, 4.0, 5.0], np.float32), len(users))
    for tag, vals in (("", np.ones(len(users), np.float32)), ("_rated", values)):
        train = sp.csr_matrix((vals, (users, items)), shape=(n_users, n_items), dtype=np.float32)
        for adj_type in ("plain", "norm", "gcmc", "mean"):
            me = types.SimpleNamespace(num_users=n_users, num_items=n_items, graph=train.toarray(),
                                       adj_type=adj_type,
                                       logger=types.SimpleNamespace(info=lambda *a: None))
            me.normalized_adj_single = types.MethodType(m["normalized_adj_single"], me)
            with np.errstate(divide="ignore"):
                adj = m["get_adj_mat"](me)
            idx, data, shape = m["_convert_sp_mat_to_sp_tensor"](me, adj)
            np.savez_compressed(os.path.join(OUT, "ngcf_adj_%s%s.npz" % (adj_type, tag)), users=users,
                                items=items, values=vals, n_users=n_users, n_items=n_items,
                                row=idx[:, 0].astype(np.int64), col=idx[:, 1].astype(np.int64),
                                data=np.asarray(data, np.float32))
            print("ngcf_adj_%s%s: nnz=%d" % (adj_type, tag, len(data)))


if __name__ == "__main__":
    main()
