"""Where one full-population evaluation spends its wall time (gowalla shape, d = 64): the whole call, and the same
call with a synchronize after each phase (tile maxima / rescoring+ranking / the rest)."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import FullRankEvaluator

train, test = synth.interactions_around_test(synth.load_test_split(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)   # the bench workload
U, I = train.shape
rng = np.random.RandomState(0)
P = torch.from_numpy(synth.xavier_uniform(U, 64, rng)).cuda()
Q = torch.from_numpy(synth.xavier_uniform(I, 64, rng)).cuda()
trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
users = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).cuda()
acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    return wrapper


for br in (32768,):
    ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=br)
    for _ in range(2):
        ev.evaluate_factors(P, Q, users)
    print("flagged rows:", ev.n_flagged)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev.evaluate_factors(P, Q, users)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("batch_rows %5d: %s ms (median %.3f)" % (br, " ".join("%.3f" % t for t in ts), sorted(ts)[3]))
    tm, et, pr = ev._gemm.tile_maxima, E.eval_tiles, ev._gemm.prepare
    ev._gemm.tile_maxima = timed("tile_maxima", tm)
    E.eval_tiles = timed("eval_tiles", et)
    ev._gemm.prepare = timed("prepare", pr)
    n = 5
    t_all = 0.0
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev.evaluate_factors(P, Q, users)
        torch.cuda.synchronize()
        t_all += (time.perf_counter() - t0) * 1e3
    print("with a sync after each phase: total %.3f ms; " % (t_all / n) + "; ".join("%s %.3f" % (k, v / n) for k, v in acc.items())
          + "; everything else %.3f" % ((t_all - sum(acc.values())) / n))
    E.eval_tiles = et
