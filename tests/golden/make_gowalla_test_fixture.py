"""The reference's real test split of gowalla (dataset/gowalla.test: 217,242 `user,item` pairs, the file
`splitter=given` reads — SURVEY §8d names it as the benchmark's test split) as a compact fixture:
tests/golden/gowalla_test_split.npz (CSR of the 29,858 x 40,981 test matrix).  gowalla.train is absent from
the reference tree, so the train side stays synthetic (neurec_amd/synth.py draws it around this split).

    python tests/golden/make_gowalla_test_fixture.py       # needs /root/reference
"""
import os

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.environ.get("NEUREC_REFERENCE", "/root/reference"), "dataset", "gowalla.test")


def main():
    pairs = np.loadtxt(SRC, delimiter=",", dtype=np.int64)
    U, I = 29858, 40981                                  # the LightGCN-paper split's sizes
    assert pairs.shape == (217242, 2) and pairs[:, 0].max() < U and pairs[:, 1].max() < I
    m = sp.csr_matrix((np.ones(len(pairs), np.int8), (pairs[:, 0], pairs[:, 1])), shape=(U, I))
    m.sum_duplicates()
    m.sort_indices()
    assert m.nnz == 217242
    np.savez_compressed(os.path.join(HERE, "gowalla_test_split.npz"), indptr=m.indptr.astype(np.int32),
                        indices=m.indices.astype(np.int32), shape=np.asarray([U, I], np.int32))
    print("users with test items:", int((np.diff(m.indptr) > 0).sum()), "pairs:", m.nnz)


if __name__ == "__main__":
    main()
