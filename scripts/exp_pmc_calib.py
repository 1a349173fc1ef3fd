"""Calibration workload for FETCH_SIZE on the SpMM access pattern (dword-per-lane row gathers):
a random graph whose gathered table (N=4M rows x 64 floats = 1 GB) exceeds L2 and Infinity Cache,
so every gathered row must come from HBM: expected fetch >= nnz*256 B + CSR stream."""
import sys, os
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E
rng = np.random.RandomState(0)
N, nnz = 4_000_000, 1_600_000
rows = np.sort(rng.randint(0, 100_000, nnz)); cols = rng.choice(N, nnz, replace=False)   # distinct rows of X
A = sp.csr_matrix((np.ones(nnz, np.float32), (rows, cols)), shape=(100_000, N)); A.sort_indices()
csr = E.SpmmCSR(A.indptr, A.indices, A.data, n_cols=N)
X = torch.randn(N, 64, device="cuda"); Y = torch.empty(100_000, 64, device="cuda")
for _ in range(5): csr.matmul(X, out=Y)
torch.cuda.synchronize()
print("nnz", A.nnz, "expected gather bytes", A.nnz * 256, "csr stream", A.nnz * 8, "Y", 100_000 * 256)
