"""The full LightGCN propagation pass (gowalla shape, `pre` adjacency) and NGCF's d = 16 pass: the lane-group SpMM vs the
vendor library behind torch.sparse.mm (rocSPARSE csrmm, fp32) on the same matrix and operand — a calibration; nothing of
it is linked into the product.  Also the max |difference| (the library's summation order is its own)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth
from neurec_amd.graph import lightgcn_adjacency, ngcf_adjacency

tr, _ = synth.interactions("gowalla", seed=2018)
U, I = tr.shape
coo = tr.tocoo()


def timed(fn, n=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, A, widths in (("lightgcn-pre", lightgcn_adjacency(coo.row, coo.col, U, I, "pre"), (64, 128)),
                        ("ngcf-norm", ngcf_adjacency(tr, "norm"), (16, 32))):
    A = A.tocsr().astype(np.float32)
    A.sort_indices()
    csr = E.SpmmCSR.from_scipy(A, split_row=U)
    At = torch.sparse_csr_tensor(torch.from_numpy(A.indptr.astype(np.int64)), torch.from_numpy(A.indices.astype(np.int64)),
                                 torch.from_numpy(A.data), size=A.shape).cuda()
    for d in widths:
        X = torch.randn(A.shape[0], d, device="cuda")
        Y = torch.empty_like(X)
        t_mine = timed(lambda: csr.matmul(X, out=Y))
        t_lib = timed(lambda: torch.sparse.mm(At, X))
        diff = float((torch.sparse.mm(At, X) - csr.matmul(X, out=Y)).abs().max())
        alg = csr.algorithmic_bytes(d)
        print("%-13s nnz %d d=%3d: this repo %6.1f us (%4.1f %% of 8 TB/s) | torch.sparse.mm %7.1f us (%4.1f %%) | max |diff| %.1e"
              % (name, A.nnz, d, t_mine, alg / t_mine / 8e4, t_lib, alg / t_lib / 8e4, diff))
