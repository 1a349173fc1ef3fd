"""GPU experiment: evaluation pipeline vs batch size (rows in flight)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import FullRankEvaluator

tr, te = synth.interactions("gowalla")
U, I = tr.shape
trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
rng = np.random.RandomState(0)
P = torch.from_numpy((rng.randn(U, 64) * 0.1).astype(np.float32)).cuda()
Q = torch.from_numpy((rng.randn(I, 64) * 0.1).astype(np.float32)).cuda()
users = torch.arange(U, dtype=torch.int32, device="cuda")
for br in (1024, 2048, 4096, 8192, 16384, 29858):
    ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=br)
    ev.evaluate_factors(P, Q, users)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = ev.evaluate_factors(P, Q, users)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("batch_rows=%5d  %.2f ms  %.2f M users/s  ndcg@10=%.6f" % (br, dt * 1e3, U / dt / 1e6, r[2 * 20 + 9]), flush=True)
    del ev; torch.cuda.empty_cache()
