"""One rank's share of a W-rank column-sharded LightGCN step at the gowalla shape, on this GPU
(what bench.py reports as colshard_one_rank_share); run under rocprofv3 for the kernel table.
    python scripts/exp_colshard_share.py W [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import _hip_timed  # noqa: E402
from neurec_amd import engine as E, parallel, synth  # noqa: E402
from neurec_amd.colshard import ColumnShardedLightGCN  # noqa: E402
from neurec_amd.graph import lightgcn_adjacency  # noqa: E402
from neurec_amd.trainer import BprEpochSampler  # noqa: E402

W = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B, dim, L = 1024, 64, 3
comm = parallel.init_from_env()
real_test = os.path.join(ROOT, "tests", "golden", "gowalla_test_split.npz")
train, test = synth.interactions_around_test(synth.load_test_split(real_test), 810128, seed=2018)
U, I = train.shape
coo = train.tocoo()
A = lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
E0 = synth.xavier_uniform(U + I, dim, np.random.RandomState(2017))
trc = E.DeviceCSR.from_scipy(train)
gB = W * B
cs = ColumnShardedLightGCN(comm, A, U, I, E0, L, 0.01, 1e-3, gB, rank=0, world=W)
sW = BprEpochSampler(trc, I, neg_num=1, batch_size=gB, shuffle=True, seed=2018, plan_users=U)
bsW = [b for b in sW.batches() if b[0].numel() == gB][:40]
itW = iter(bsW * 8)


def one_share():
    b = next(itW)
    cs.step(b[0], b[1], b[2], None, plan=b.plan)


ms = _hip_timed(one_share, steps, 10)
print("W=%d global batch %d columns %d kernel width %d: %.4f ms/step" % (W, gB, dim // W, cs.local.d, ms))
