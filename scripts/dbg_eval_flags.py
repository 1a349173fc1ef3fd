import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, scipy.sparse as sp
import test_eval_gpu as T
from neurec_amd import engine as E
from neurec_amd.trainer import FullRankEvaluator
d, kind, extra = 48, "underflow-edge", 2
rng = np.random.RandomState(d + 5)
U, I = 600, 6000
P = (rng.randn(U, d) * 0.1).astype(np.float32); Q = (rng.randn(I, d) * 0.1).astype(np.float32)
P, Q = T._spread_tables(rng, U, I, d, kind)
tr = sp.random(U, I, 0.01, random_state=1, format="lil", dtype=np.float32)
for u in range(0, U, 3):
    tr[u, np.argsort(-(P[u] @ Q.T))[:rng.randint(1, 30)]] = 1.0
tr = tr.tocsr(); tr.data[:] = 1.0; tr.sort_indices()
te = sp.random(U, I, 0.004, random_state=2, format="csr", dtype=np.float32)
te = te - te.multiply(tr); te.eliminate_zeros(); te.sort_indices()
users = np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)
trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda(); ud = torch.from_numpy(users).cuda()
for native in ("1", "0"):
    os.environ["NEUREC_EVAL_NATIVE_LOOP"] = native
    res = {}
    for name, kw in (("full", dict(pruned=False)), ("fp32", dict(search="fp32")), ("bf16", dict(search="bf16", extra_tiles=extra))):
        ev = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=256, **kw)
        rows = ev.evaluate_factors(Pd, Qd, ud, per_user=True)
        nf1 = ev.n_flagged
        m1 = ev.evaluate_factors(Pd, Qd, ud)
        nf2 = ev.n_flagged
        m2 = ev.evaluate_factors(Pd, Qd, ud)
        res[name] = (rows, m1, m2)
        print(native, name, "flagged", nf1, nf2, "means equal across calls", np.array_equal(m1, m2), "mean vs rows", np.abs(m1 - rows.astype(np.float64).mean(0)).max())
    for k in ("fp32", "bf16"):
        print(native, k, "rows equal full:", np.array_equal(res[k][0], res["full"][0]), "means equal full:", np.array_equal(res[k][1], res["full"][1]), np.abs(res[k][1]-res["full"][1]).max())
