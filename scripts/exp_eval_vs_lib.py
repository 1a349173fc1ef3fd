"""Full-rank evaluation of the gowalla population (29,858 users x 40,981 items, d = 64, K = 20): this repo's pruned
evaluator (scores -> strikes -> top-K -> five metrics, one host copy) vs the straightforward torch formulation of its
first three stages only (torch.mm -> -inf on the train items -> torch.topk; no metrics) on the same tables — a calibration."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import FullRankEvaluator

train, test = synth.interactions("gowalla", seed=2018)
U, I = train.shape
rng = np.random.RandomState(0)
P = torch.from_numpy(synth.xavier_uniform(U, 64, rng)).cuda()
Q = torch.from_numpy(synth.xavier_uniform(I, 64, rng)).cuda()
trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
users = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).cuda()
coo = train.tocoo()
tr_rows = torch.from_numpy(coo.row.astype(np.int64)).cuda()
tr_cols = torch.from_numpy(coo.col.astype(np.int64)).cuda()
row_of = torch.full((U,), -1, dtype=torch.int64, device="cuda")
row_of[users.long()] = torch.arange(users.numel(), device="cuda")
torch.backends.cuda.matmul.allow_tf32 = False


def wall(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[n // 2]


ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=32768)
t_mine = wall(lambda: ev.evaluate_factors(P, Q, users))


def torch_way(batch=8192):
    out = []
    for b in range(0, users.numel(), batch):
        u = users[b:b + batch].long()
        S = torch.mm(P[u], Q.t())
        r = row_of[tr_rows] - b                                   # train pairs of the batch's users -> -inf
        m = (r >= 0) & (r < u.numel())
        S[r[m], tr_cols[m]] = float("-inf")
        out.append(torch.topk(S, 20, dim=1).indices)
    return torch.cat(out)


t_torch = wall(torch_way)
top = torch_way()
print("this repo, whole evaluation (scores, strikes, top-20, 5 metrics x 20 cut-offs, host copy): %.2f ms = %.1f M users/s"
      % (t_mine, users.numel() / t_mine / 1e3))
print("torch.mm + index_put(-inf) + torch.topk (no metrics), batches of 8192:                     %.2f ms = %.1f M users/s"
      % (t_torch, users.numel() / t_torch / 1e3))
