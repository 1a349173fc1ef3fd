"""GPU experiment: full-rank evaluation with the ranking of batch b overlapped with the scoring of
batch b+1 (two streams, two slabs) vs serial, by batch size; results must be identical."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import FullRankEvaluator

tr, te = synth.interactions("gowalla")
U, I = tr.shape
rng = np.random.RandomState(0)
P = torch.from_numpy((rng.randn(U, 64) * 0.1).astype(np.float32)).cuda()
Q = torch.from_numpy((rng.randn(I, 64) * 0.1).astype(np.float32)).cuda()
trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
users = torch.from_numpy(np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)).cuda()
ref = None
for rows in (4096, 8192, 16384, 29858):
    for overlap, pruned in ((False, False), (True, False), (False, True)):
        ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=rows, overlap=overlap, pruned=pruned)
        m = ev.evaluate_factors(P, Q, users)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m = ev.evaluate_factors(P, Q, users)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        if ref is None:
            ref = m
        print("batch_rows=%5d overlap=%d pruned=%d : %.2f ms  %.2f M users/s  same=%s flagged=%s"
              % (rows, overlap, pruned, dt * 1e3, users.numel() / dt / 1e6, np.array_equal(ref, m),
                 getattr(ev, "n_flagged", "-")), flush=True)
