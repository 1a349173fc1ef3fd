"""Does a HIP graph shorten one full-population evaluation (gowalla shape, d = 64)?  The evaluation's ~17 launches
(two Python-level prepare calls + nrhip_eval_pruned) issued directly, against the same launches captured once into a
graph (torch.cuda.CUDAGraph around the ctypes launches: they go to torch's current stream) and replayed."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import FullRankEvaluator

train, test = synth.interactions_around_test(synth.load_test_split(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
U, I = train.shape
rng = np.random.RandomState(0)
P = torch.from_numpy(synth.xavier_uniform(U, 64, rng)).cuda()
Q = torch.from_numpy(synth.xavier_uniform(I, 64, rng)).cuda()
trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
users = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).cuda()
ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=32768)
for _ in range(3):
    want = ev.evaluate_factors(P, Q, users)
n = users.numel()
per_user = torch.empty((n, 5 * 20), dtype=torch.float32, device="cuda")
flags = torch.empty(n, dtype=torch.int32, device="cuda")


def body():
    ev._gemm.prepare(Q)
    ev._filter.prepare(Q)
    return ev._native.run(P, Q, users, ev._row_of, per_user, flags, prepare_items=False)


def med(fn, reps=15):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], min(ts)


body()
torch.cuda.synchronize()
print("direct launches : median %.3f ms, best %.3f" % med(body))
direct = ev._native.sums.cpu().numpy().copy()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    body()
torch.cuda.synchronize()
ev._native.sums.zero_()
print("graph replay    : median %.3f ms, best %.3f" % med(g.replay))
replayed = ev._native.sums.cpu().numpy()
print("sums equal:", bool(np.array_equal(direct, replayed)), " ndcg@10 %.8f" % (direct[2 * 20 + 9] / n))
