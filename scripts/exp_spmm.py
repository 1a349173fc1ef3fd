"""GPU experiment: SpMM time vs size of the gathered table (is the gather L2- or MALL-bound?)."""
import sys, os, time
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E

def bench(csr, X, reps=30, **kw):
    Y = torch.empty_like(X)
    for _ in range(3): csr.matmul(X, out=Y, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): csr.matmul(X, out=Y, **kw)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

rng = np.random.RandomState(0)
nnz = 1_630_000
print("N, d, table_MB, us, gather_TB/s")
for d in (16, 32, 64, 128):
    for N in (2000, 8000, 16000, 35000, 70839, 150000, 300000, 1200000):
        rows = np.sort(rng.randint(0, N, nnz)); cols = rng.randint(0, N, nnz)
        A = sp.csr_matrix((np.ones(nnz, np.float32), (rows, cols)), shape=(N, N)); A.sum_duplicates(); A.sort_indices()
        csr = E.SpmmCSR.from_scipy(A)
        X = torch.randn(N, d, device="cuda")
        us = bench(csr, X)
        print(N, d, round(N * d * 4 / 1e6, 1), round(us, 1), round(A.nnz * d * 4 / us / 1e6, 2), flush=True)
