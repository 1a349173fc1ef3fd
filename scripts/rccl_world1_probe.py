import os, sys, json, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577",
                  NEUREC_DIST_BACKEND="nccl", NEUREC_DIST_FORCE_GROUP="1")
import torch, numpy as np
t0=time.time()
from neurec_amd import parallel, graph
import test_rccl_gpu as T
comm = parallel.init_from_env()
print("init", time.time()-t0, comm.live, comm.backend, flush=True)
T._comm_methods(comm, torch)
print("comm methods ok", comm.calls, time.time()-t0, flush=True)
from neurec_amd.sharded import ShardedLightGCN
tr, coo, E0, U, I = T._graph(64)
A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
res=[]
for c, pipe in ((comm, True), (comm, False), (parallel.Comm(), True), (parallel.Comm(), False)):
    eng = ShardedLightGCN(c, A, U, I, E0, 3, 0.01, 1e-3, 128, pipeline=pipe)
    for b in T._batches(U, I, 128, 3):
        eng.step(*(torch.from_numpy(x).cuda() for x in b), torch.zeros(2, device="cuda"))
    res.append(eng.E0.cpu().numpy())
print("equal:", [np.array_equal(res[0], r) for r in res], comm.calls, flush=True)
comm.shutdown()
