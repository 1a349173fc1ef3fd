"""Randomised bit-identity check of the BPR-MF step's lazy forms (one launch / two launches / the native batch loop)
against TF's literal all-rows sweep: tables and both moments after a run of steps, over table sizes, widths, batch sizes
(1 ... 9,000), replay periods, hot rows and rows never touched."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd.trainer import MFEngine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
for case in range(n_cases):
    U = int(rng.choice([30, 700, 5000, 40000])); I = int(rng.choice([50, 900, 6000, 60000]))
    d = int(rng.choice([8, 16, 20, 64, 100, 128])); B = int(rng.choice([1, 7, 256, 512, 2048, 9000]))
    period = int(rng.choice([1, 3, 8, 16, 200])); reg = float(rng.choice([0.0, 0.01])); steps = int(rng.choice([5, 40, 90]))
    P0 = (rng.randn(U, d) * 0.05).astype(np.float32); Q0 = (rng.randn(I, d) * 0.05).astype(np.float32)
    hot_u, hot_i = np.arange(min(20, U)), np.arange(min(30, I))
    batches = []
    for s in range(steps):
        b = B if s % 7 != 6 else max(1, B // 3)
        pick = lambda hot, n: np.where(rng.rand(b) < 0.6, rng.choice(hot, b), rng.randint(0, max(1, int(n * 0.9)), b)).astype(np.int32)
        batches.append((pick(hot_u, U), pick(hot_i, I), pick(hot_i, I)))
    sweep = MFEngine(P0, Q0, 0.003, reg, B, lazy=False)
    lb = torch.zeros(2, device="cuda")
    losses = []
    for bu, bp, bn in batches:
        sweep.step(dev(bu), dev(bp), dev(bn), lb)
        losses.append(lb.cpu().numpy().copy())
    want = {k: getattr(sweep, k).cpu().numpy() for k in ("P", "Q", "mP", "mQ", "vP", "vQ")}
    for form in ("one-launch", "two-launch", "native loop"):
        eng = MFEngine(P0, Q0, 0.003, reg, B, lazy=True, lazy_period=period, fused=form != "two-launch")
        la = torch.zeros(2, device="cuda")
        ok = True
        if form == "native loop":
            # the batch loop in C over a stream cut into equal batches (the last one short)
            n = sum(len(x[0]) for x in batches)
            us = np.concatenate([x[0] for x in batches]); ps = np.concatenate([x[1] for x in batches]); ns = np.concatenate([x[2] for x in batches])
            sw2 = MFEngine(P0, Q0, 0.003, reg, B, lazy=False)
            k = (n + B - 1) // B
            for s in range(k):
                sw2.step(dev(us[s * B:(s + 1) * B]), dev(ps[s * B:(s + 1) * B]), dev(ns[s * B:(s + 1) * B]), lb)
            ref = {kk: getattr(sw2, kk).cpu().numpy() for kk in want}
            ls = torch.zeros(2 * k, device="cuda")
            eng.run_batches(dev(us), dev(ps), dev(ns), B, ls)
        else:
            ref = want
            for (bu, bp, bn), l in zip(batches, losses):
                eng.step(dev(bu), dev(bp), dev(bn), la)
                g = la.cpu().numpy()
                ok = ok and g[0] == l[0] and g[1] == l[1]
        for kk in ref:
            ok = ok and np.array_equal(getattr(eng, kk).cpu().numpy().view(np.uint32), ref[kk].view(np.uint32))
        if not ok:
            bad += 1
            print("MISMATCH (%s)" % form, end=" ")
    print("case %2d: U=%d I=%d d=%d B=%d period=%d reg=%g steps=%d" % (case, U, I, d, B, period, reg, steps), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
