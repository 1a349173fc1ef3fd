"""GPU driver for counter passes: 20 plain lane-group SpMM passes (d = 64) on the bench's gowalla-shaped
graph (synth.interactions_around_test on the committed test split, as bench.py builds it)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth, graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tr, te = synth.interactions_around_test(
    synth.load_test_split(os.path.join(ROOT, "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
coo = tr.tocoo(); U, I = tr.shape
A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
X = torch.randn(U + I, 64, device="cuda")
csr = E.SpmmCSR.from_scipy(A, split_row=U)
Y = torch.empty_like(X)
for _ in range(20):
    csr.matmul(X, out=Y)
torch.cuda.synchronize()
