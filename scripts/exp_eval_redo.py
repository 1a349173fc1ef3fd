"""What redone rows cost (gowalla shape, d = 64): the clean evaluation against the same evaluation with 0.1 % / 1 % of
its rows redone from full fp32 score rows (nrhip_eval_redo).  Two causes: (a) rows marked uncertified by hand after the
search (the flag an int8 / bf16 bound that did not hold sets: ordinary rows, ranked again by the streaming selection),
(b) NaN factor rows (nothing comparable in the row: flagged by the search itself, and the redo's exact tie path replays
the reference's heap over all 40,981 columns — the slowest row there is)."""
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
P0 = synth.xavier_uniform(U, 64, rng)
Q = torch.from_numpy(synth.xavier_uniform(I, 64, rng)).cuda()
trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
users = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).cuda()
n = users.numel()


def med(ev, P, reps=9):
    for _ in range(3):
        ev.evaluate_factors(P, Q, users)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev.evaluate_factors(P, Q, users)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


for search in ("int8", "bf16"):
    base = None
    for frac in (0.0, 0.001, 0.01):
        k = int(round(frac * n))
        ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=32768, search=search)
        ev.int8_retry = 0                                        # (keep the search under test: no pause after uncertified rows)
        if k:
            marked = torch.from_numpy(np.random.RandomState(1).choice(n, k, replace=False)).cuda()
            read = ev._read_native_sums

            def patched(read=read, ev=ev, marked=marked, state={"first": True}):
                both = read()
                if state["first"]:                               # behind the search: mark the rows, report the count
                    ev._flags_buf[marked] = 2
                    both[-2] += marked.numel(); both[-1] += marked.numel()
                state["first"] = not state["first"]              # (the second read of an evaluation: the retaken sums)
                return both
            ev._read_native_sums = patched
        t = med(ev, torch.from_numpy(P0).cuda())
        base = base or t
        print("%s search, %5.1f %% of the rows marked uncertified: %4d rows redone, %.3f ms per evaluation (%.2f x the clean one)"
              % (search, 100 * frac, ev.n_flagged, t, t / base))
for frac in (0.001, 0.01):
    P = P0.copy()
    P[np.random.RandomState(1).choice(U, int(round(frac * U)), replace=False)] = np.nan
    ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=32768, search="bf16")
    t = med(ev, torch.from_numpy(P).cuda())
    print("bf16 search, %5.1f %% of the user rows NaN: %4d rows redone (exact tie path), %.3f ms per evaluation" % (100 * frac, ev.n_flagged, t))
