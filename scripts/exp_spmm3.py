"""GPU experiment: SpMM tuning sweep on the gowalla-shaped graph (item rows/nnz; gathers via env)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth, graph

def bench(fn, reps=40):
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
d = int(os.environ.get("D", "64"))
X = torch.randn(U + I, d, device="cuda"); Y = torch.empty_like(X); S = torch.randn_like(X)
g = os.environ.get("NRHIP_SPMM_GATHER", "16")
for rows in (8, 16, 32):
    for nnz in (64, 128, 256):
        csr = E.SpmmCSR.from_scipy(A, item_rows=rows, item_nnz=nnz)
        us = bench(lambda: csr.matmul(X, out=Y, sum_in=S, sum_out=S))
        print("G=%s d=%d rows=%2d nnz=%3d items=%6d : %.1f us" % (g, d, rows, nnz, csr.n_segments, us), flush=True)
