"""BPR-MF tables (config 2 after a few hundred steps) through the evaluator under each search arithmetic: time of each
of 8 consecutive evaluations, which arithmetic ran, how many rows were redone."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import FullRankEvaluator

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
train, test = synth.interactions_around_test(synth.load_test_split(os.path.join(ROOT, "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
U, I = train.shape
rng = np.random.RandomState(0)
trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
users = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).cuda()
for name, scale in (("N(0, 0.01) (MF init)", 0.01), ("N(0, 0.1)", 0.1)):
    P = torch.from_numpy((rng.randn(U, 64) * scale).astype(np.float32)).cuda()
    Q = torch.from_numpy((rng.randn(I, 64) * scale).astype(np.float32)).cuda()
    for search, extra in (("int8", 0), ("int8", 2), ("int8", 4), ("int8", 6), ("int8", 8), ("bf16", 0)):
        ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=32768, search=search)
        ev.int8_retry, ev.int8_extra_tiles = 0, extra
        out = ["+%d tiles" % extra]
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ev.evaluate_factors(P, Q, users)
            torch.cuda.synchronize()
            out.append("%.3f/%s/%d" % ((time.perf_counter() - t0) * 1e3, ev.search_used, ev.n_flagged))
        print(name, search, " ".join(out))
