"""Full-pass SpMM at the gowalla shape under the affinity-schedule knobs (A/B on the GPU box):
    NEUREC_SPMM_AFFINITY=0|1   NEUREC_SPMM_AFF_BLOCK=<window bytes>
prints us per pass (HIP events, 200 passes)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth
from neurec_amd.graph import lightgcn_adjacency

tr, _ = synth.interactions("gowalla", seed=2018)
U, I = tr.shape
coo = tr.tocoo()
A = lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
csr = E.SpmmCSR.from_scipy(A, split_row=U)
X = torch.from_numpy(synth.xavier_uniform(U + I, 64, np.random.RandomState(1))).cuda()
Y = torch.empty_like(X)
csr.matmul(X, out=Y)
torch.cuda.synchronize()
wb = E.C.c_int(0)
wa = E._lib.lib.nrhip_spmm_blocked_affinity(csr.blocked, E.C.byref(wb)) if csr.blocked else -1
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(2):
    ev0.record()
    for _ in range(200):
        csr.matmul(X, out=Y)
    ev1.record()
    torch.cuda.synchronize()
print("AFFINITY=%s BLOCK=%s windows=(%d,%d): %.2f us/pass" % (
    os.environ.get("NEUREC_SPMM_AFFINITY", "1"), os.environ.get("NEUREC_SPMM_AFF_BLOCK", "default"),
    wa, wb.value, ev0.elapsed_time(ev1) * 1000 / 200))
