"""What the column-sharded step's ONE collective costs the loop when it really goes through RCCL: the engine at world
size 1 with a forced process group (backend nccl) against the same engine without one.  (One rank: the all-gather moves
nothing; what shows is the issue cost of all_gather_into_tensor + work.wait() per step and whether the loop stays
GPU-bound.)"""
import os, sys, time
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29591",
                  NEUREC_DIST_BACKEND="nccl", NEUREC_DIST_FORCE_GROUP="1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neurec_amd import engine as E, parallel, synth
from neurec_amd.colshard import ColumnShardedLightGCN
from neurec_amd.graph import lightgcn_adjacency
from neurec_amd.trainer import BprEpochSampler
comm = parallel.init_from_env()
train, _ = synth.interactions_around_test(synth.load_test_split(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
U, I = train.shape
coo = train.tocoo()
A = lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
E0 = synth.xavier_uniform(U + I, 64, np.random.RandomState(2017))
trc = E.DeviceCSR.from_scipy(train)
for name, c in (("no process group", parallel.Comm()), ("RCCL, world 1 (forced)", comm)):
    for W in (1, 8):
        eng = ColumnShardedLightGCN(c, A, U, I, E0, 3, 0.01, 1e-3, W * 1024, rank=0, world=W) if W > 1 else \
            ColumnShardedLightGCN(c, A, U, I, E0, 3, 0.01, 1e-3, 1024)
        sam = BprEpochSampler(trc, I, neg_num=1, batch_size=W * 1024, shuffle=True, seed=2018, plan_users=U)
        bs = [b for b in sam.batches() if b[0].numel() == W * 1024][:50]
        for b in bs[:10]:
            eng.step(b[0], b[1], b[2], None, plan=b.plan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            for b in bs:
                eng.step(b[0], b[1], b[2], None, plan=b.plan)
        t_issue = (time.perf_counter() - t0) / 200
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / 200
        print("%-24s columns of %d rank(s): %.1f us per step (host issue %.1f us)%s"
              % (name, W, t_all * 1e6, t_issue * 1e6, "   [one rank's share, no collective]" if W > 1 else ""))
comm.shutdown()
