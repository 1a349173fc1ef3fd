"""20 launches of the lane-group SpMM at d = 16 (NGCF norm adjacency, gowalla shape) — for counter passes"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth
from neurec_amd.graph import ngcf_adjacency
tr, _ = synth.interactions("gowalla", seed=2018)
A = ngcf_adjacency(tr, "norm")
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
csr = E.SpmmCSR.from_scipy(A, split_row=tr.shape[0])
X = torch.randn(A.shape[0], d, device="cuda")
Y = torch.empty_like(X)
for _ in range(20):
    csr.matmul(X, out=Y)
torch.cuda.synchronize()
