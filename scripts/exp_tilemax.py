"""Evaluation level 1 at the gowalla shape: in-loop strikes vs unmasked loop + planned fix-up (HIP events)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth

train, test = synth.interactions("gowalla", seed=2018)
U, I = train.shape
d = 64
rng = np.random.RandomState(0)
P = torch.from_numpy((rng.randn(U, d) * 0.1).astype(np.float32)).cuda()
Q = torch.from_numpy((rng.randn(I, d) * 0.1).astype(np.float32)).cuda()
trc = E.DeviceCSR.from_scipy(train)
users = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).cuda()
n = users.numel()
BR = 16384
gemm = E.ScoreGemm(Q, BR)
plan = E.TileStrikePlan(trc, I)
print("plan: %d pairs, %d chunks" % (plan.n_pairs, plan.n_chunks))
row_of = torch.full((U,), -1, dtype=torch.int32, device="cuda")
row_of[users.long()] = torch.arange(n, dtype=torch.int32, device="cuda")
out = torch.empty((BR, 1284), dtype=torch.float32, device="cuda")


def run(pl):
    for b in range(0, n, BR):
        gemm.tile_maxima(P, users[b:b + BR], trc, out=out, plan=pl, row_of=row_of if pl else None, row_lo=b)


for name, pl in (("in-loop strikes", None), ("unmasked + planned fix-up", plan)):
    for _ in range(2):
        run(pl)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run(pl)
    b.record()
    torch.cuda.synchronize()
    print("%-28s %.1f us per evaluation of %d users" % (name, a.elapsed_time(b) / 5 * 1e3, n))
