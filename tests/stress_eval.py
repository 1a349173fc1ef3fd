"""Randomised cross-check of the pruned evaluation against the materialised path (same per-user metric rows, bit for bit)
over shapes, cut-offs, searches, batch sizes and awkward rows (zero / NaN / coarse-valued tables that tie everywhere)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E
from neurec_amd.trainer import FullRankEvaluator
from oracle.native import lists_to_csr

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    U = int(rng.choice([37, 200, 513, 1500]))
    I = int(rng.choice([90, 700, 2049, 5000, 33000, 70001]))
    d = int(rng.choice([8, 16, 20, 32, 50, 64, 100, 128]))
    top_k = int(rng.choice([1, 3, 10, 20, 33, 50, 62]))
    if top_k > I // 4:
        top_k = max(1, I // 8)
    kind = rng.choice(["gauss", "coarse", "zero_rows", "nan_rows", "popular", "nan_items", "inf_items", "huge", "tiny", "dense_train", "repeat_users"])
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    if kind == "coarse":
        P, Q = np.round(P * 20) / 20, np.round(Q * 20) / 20
    if kind == "zero_rows":
        P[rng.rand(U) < 0.2] = 0
    if kind == "nan_rows":
        P[rng.rand(U) < 0.1] = np.nan
    if kind == "nan_items":
        Q[rng.choice(I, 3, replace=False)] = np.nan
    if kind == "inf_items":
        Q[rng.choice(I, 2, replace=False), 0] = np.inf
    if kind == "huge":
        P *= 1e18; Q *= 1e18                                     # products overflow to inf for some pairs
    if kind == "tiny":
        P *= 1e-22; Q *= 1e-22                                   # products are sub-normal or flush to zero
    if kind == "popular":
        Q[rng.choice(I, min(40, I), replace=False)] += (0.3 * np.sign(P.mean(0) + 1e-3)).astype(np.float32)
    tr, te = [], []
    for u in range(U):
        a = set(rng.randint(0, I, rng.randint(0, 40)).tolist())
        b = set(rng.randint(0, I, rng.randint(0, 8)).tolist()) - a
        if kind == "dense_train" and rng.rand() < 0.15:         # almost every item struck: the rest of the ranking is -inf ties
            keep = set(rng.choice(I, min(I, int(rng.choice([0, 3, 25]))), replace=False).tolist())
            a = set(range(I)) - keep
            b = set(list(keep)[:2])
        tr.append(sorted(a)); te.append(sorted(b))
    def csr(lists):
        ptr, idx = lists_to_csr(lists)
        return E.DeviceCSR(ptr, idx[:max(int(ptr[-1]), 1)], I)
    trc, tec = csr(tr), csr(te)
    ulist = [u for u in range(U) if te[u]]
    if kind == "repeat_users" and len(ulist) > 3:
        ulist = ulist + ulist[:3] + [ulist[1]]                   # a user list with repeats (the in-loop strikes' case)
    users = torch.from_numpy(np.asarray(ulist, np.int32)).cuda()
    if users.numel() == 0:
        continue
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    br = int(rng.choice([64, 256, 4096]))
    want = np.asarray(FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], top_k, batch_rows=br, pruned=False)
                      .evaluate_factors(Pd, Qd, users, per_user=True))
    for search in ("int8", "bf16", "fp32"):
        ev = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], top_k, batch_rows=br, search=search,
                               extra_tiles=int(rng.choice([0, 2, 8, 30])))
        for rep in range(2):
            got = np.asarray(ev.evaluate_factors(Pd, Qd, users, per_user=True))
            ok = np.array_equal(got.view(np.uint32), want.view(np.uint32))
            sums = ev.evaluate_factors(Pd, Qd, users, column_sums=True)
            ref = want.astype(np.float64).sum(0)
            ok2 = np.allclose(sums, ref, rtol=1e-12, atol=0, equal_nan=True)    # (NaN factor rows: NaN metrics on both sides)
            if not (ok and ok2):
                rows_bad = np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(1))
                nan_users = np.isnan(P[users.cpu().numpy()]).any(1)
                bad += 1
                if rows_bad.size:
                    r0 = int(rows_bad[0]); u0 = int(users[r0])
                    print("  first bad row %d (user %d, nan user: %s, flagged rows %d): got %s want %s; bad rows %s; nan flags of bad rows %s"
                          % (r0, u0, bool(nan_users[r0]), ev.n_flagged, got[r0][:6], want[r0][:6], rows_bad[:10], nan_users[rows_bad][:10]), flush=True)
                print("MISMATCH case %d: U=%d I=%d d=%d K=%d %s batch=%d search=%s rep=%d rows_equal=%s sums_equal=%s flagged=%d"
                      % (case, U, I, d, top_k, kind, br, search, rep, ok, ok2, ev.n_flagged), flush=True)
    print("case %2d ok: U=%d I=%d d=%d K=%d %s batch=%d" % (case, U, I, d, top_k, kind, br), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
