"""Where the config-4 propagation pass spends its gathers: the two halves of the bipartite pass
timed on their own at FULL size on one GPU —
    user rows  (10^7 rows, 2·10^8 non-zeros) gather ITEM rows  (10^6 x 512 B = 0.5 GB, Zipf(0.8) popularity)
    item rows  (10^6 rows, 2·10^8 non-zeros) gather USER rows  (10^7 x 512 B = 5.1 GB, every row ~20 times, never close in time)
with the reduce hop's matrices (sharded.ShardedLightGCN(hop="reduce") at one rank: Mu = my user rows against
[own users ; all items], Mp = every item row against my user rows).  Run under rocprofv3 --pmc FETCH_SIZE for the
L2 -> fabric bytes of each kernel (scripts/r06_config4_halves.sh).  Prints one JSON line."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from neurec_amd import engine as E, parallel as par, synth
from neurec_amd.sharded import ShardedLightGCN

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dim = 128
dev = torch.device("cuda", 0)
U, I, n_edges = (max(int(x * scale), 64) for x in synth.CONFIG4)
tr_ptr, tr_idx = synth.device_interactions(U, I, n_edges, seed=2018, device=dev)
comm = par.Comm()
rows = synth.device_lightgcn_rank_rows(tr_ptr, tr_idx, U, I, (0, U), (0, I))
lim = float(np.sqrt(6.0 / (U + I + dim)))
E0 = (torch.rand(U + I, dim, device=dev) * 2 - 1) * lim
lg = ShardedLightGCN(comm, None, U, I, E0, 3, 0.01, 1e-3, 8192, local_rows=rows, hop="reduce")
del rows, E0
Mu, Mp = lg.R
bu = lg.part.bu
lg._Z.uniform_(-lim, lim)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t_users = timed(lambda: Mu.matmul(lg._Z, out=lg.Ya[0][:bu]))          # user rows <- item rows
t_items = timed(lambda: Mp.matmul(lg.E0[0], out=lg._P))               # item rows <- user rows
t_full = timed(lambda: lg.A.matmul(lg.E0[0], out=lg.Yb[0]))           # the whole pass, one launch
row = dim * 4
out = {"scale": scale, "users": U, "items": I, "dim": dim,
       "user_rows": {"ms": t_users, "nnz": int(Mu.nnz), "gather_GBps": Mu.nnz * row / t_users / 1e6,
                     "gathered_table_MB": I * row / 1e6, "kernel": Mu.full_pass_kernel(dim)},
       "item_rows": {"ms": t_items, "nnz": int(Mp.nnz), "gather_GBps": Mp.nnz * row / t_items / 1e6,
                     "gathered_table_MB": U * row / 1e6, "kernel": Mp.full_pass_kernel(dim)},
       "whole_pass": {"ms": t_full, "nnz": int(lg.A.nnz), "gather_GBps": lg.A.nnz * row / t_full / 1e6}}
print(json.dumps(out))
