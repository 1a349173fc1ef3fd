"""GPU experiment: SpMM on the gowalla-shaped graph (forward/backward variants) + synthetic sizes."""
import sys, os
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth, graph

def bench(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

tr, te = synth.interactions("gowalla")
coo = tr.tocoo(); U, I = tr.shape
A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
csr = E.SpmmCSR.from_scipy(A)
print("items", csr.n_segments, "split rows", csr.n_split_rows, "nnz", csr.nnz)
for d in (64, 128):
    X = torch.randn(U + I, d, device="cuda"); Y = torch.empty_like(X); H = torch.randn_like(X); S = torch.randn_like(X)
    alg = csr.algorithmic_bytes(d)
    for name, fn in [("plain", lambda: csr.matmul(X, out=Y)),
                     ("fwd(sum)", lambda: csr.matmul(X, out=Y, sum_in=S, sum_out=S)),
                     ("fwd(last)", lambda: csr.matmul(X, out=None, sum_in=S, sum_out=S)),
                     ("bwd(addend)", lambda: csr.matmul(X, out=Y, addend=H))]:
        us = bench(fn)
        print("gowalla d=%d %-12s %.1f us  alg %.0f GB/s  gather %.2f TB/s" % (d, name, us, alg / us / 1e3, csr.nnz * d * 4 / us / 1e6), flush=True)
rng = np.random.RandomState(0)
nnz = 1_630_000
for N in (8000, 70839, 300000):
    rows = np.sort(rng.randint(0, N, nnz)); cols = rng.randint(0, N, nnz)
    M = sp.csr_matrix((np.ones(nnz, np.float32), (rows, cols)), shape=(N, N)); M.sum_duplicates(); M.sort_indices()
    c = E.SpmmCSR.from_scipy(M); X = torch.randn(N, 64, device="cuda"); Y = torch.empty_like(X)
    us = bench(lambda: c.matmul(X, out=Y))
    print("random N=%d d=64: %.1f us gather %.2f TB/s" % (N, us, M.nnz * 256 / us / 1e6), flush=True)
