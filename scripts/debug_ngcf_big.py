"""debug: where does the NGCF engine's E0 differ from oracle.train at the gowalla shape?"""
import numpy as np, torch, sys
sys.path.insert(0, ".")
from neurec_amd import synth
from neurec_amd.graph import ngcf_adjacency, transpose_csr
from neurec_amd.trainer import NGCFEngine
from oracle import train as O
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
tr, _ = synth.interactions("gowalla", seed=2018)
U, I = tr.shape
A = ngcf_adjacency(tr, "norm"); At = transpose_csr(A)
rng = np.random.RandomState(2017)
d, B, lr, reg, drop = 16, 512, 0.001, 0.0, 0.1
E0 = synth.xavier_uniform(U + I, d, rng)
W = [tuple((rng.randn(*s) * np.sqrt(1.3 * 2 / (s[0] + s[1]))).astype(np.float32) for s in ((d, d), (1, d), (d, d), (1, d))) for _ in range(2)]
coo = tr.tocoo()
pick = rng.randint(0, coo.nnz, B)
bu, bp, bn = coo.row[pick].astype(np.int32), coo.col[pick].astype(np.int32), rng.randint(0, I, B).astype(np.int32)
masks = [(rng.rand(U + I, d) < 1 - drop).astype(np.uint8) for _ in W]
eng = NGCFEngine(A, At, U, I, E0, W, lr, reg, drop, B)
loss2 = torch.zeros(2, device="cuda")
eng.step(dev(bu), dev(bp), dev(bn), loss2, masks=[dev(m) for m in masks])
got = eng.E0.cpu().numpy()
res = {}
for dt in (np.float32, np.float64):
    A_, At_ = A.astype(dt), At.astype(dt)
    oE = E0.astype(dt); oW = [[w.astype(dt) for w in ws] for ws in W]
    loss, dE, wg = O.ngcf_loss_and_grads(A_, At_, oE, [tuple(ws) for ws in oW], [m.astype(dt) for m in masks], 1 - drop, U, bu, bp, bn, reg)
    m, v = np.zeros_like(oE), np.zeros_like(oE)
    O.Adam(lr, dtype=dt).dense(oE, m, v, dE)
    res[dt] = (oE, dE)
e32, g32 = res[np.float32]; e64, g64 = res[np.float64]
err = np.abs(got - e64)
print("after ONE step: max err vs fp64 %.2e, vs fp32 %.2e, oracle bar %.2e" % (err.max(), np.abs(got - e32).max(), np.abs(e32 - e64).max()))
deg = np.diff(A.indptr); degt = np.diff(At.indptr)
idx = np.argsort(-err.ravel())[:12]
for f in idx:
    r, c = divmod(int(f), d)
    print("row %6d (%s) col %2d deg(A) %5d deg(At) %5d  got-E0 %+.3e  o32-E0 %+.3e  o64-E0 %+.3e  g32 %+.3e g64 %+.3e  inbatch=%s" % (
        r, "user" if r < U else "item", c, deg[r], degt[r], got[r, c] - E0[r, c], e32[r, c] - E0[r, c], e64[r, c] - E0[r, c], g32[r, c], g64[r, c],
        bool((bu == r).any() or (bp + U == r).any() or (bn + U == r).any())))
# implied gradient from the engine's update (step 1: update = alpha*0.1*g/(sqrt(0.001)*|g|+eps))
gE = eng.gE0.cpu().numpy()
print("gE0 vs g64: max abs %.3e ; vs g32 %.3e ; g32 vs g64 %.3e ; |g64| max %.3e" % (np.abs(gE - g64).max(), np.abs(gE - g32).max(), np.abs(g32 - g64).max(), np.abs(g64).max()))
e = np.abs(gE - g64)
r, c = np.unravel_index(np.argmax(e), e.shape)
print("worst gradient coord row %d col %d: got %.6e g32 %.6e g64 %.6e deg %d inbatch=%s" % (r, c, gE[r, c], g32[r, c], g64[r, c], deg[r], bool((bu == r).any() or (bp + U == r).any() or (bn + U == r).any())))
# the last SpMM of the backward: nxt = dEd + At @ dS with the ENGINE's own dS / dEd (layer 0 leaves them)
dS, dEd = eng.dS.cpu().numpy().astype(np.float64), eng.dEd.cpu().numpy().astype(np.float64)
nxt = eng.dEgo[0].cpu().numpy()
want = dEd + At.astype(np.float64) @ dS
print("last hop (engine inputs, fp64 host product): max abs err %.3e, |want| max %.3e" % (np.abs(nxt - want).max(), np.abs(want).max()))
ee = np.abs(nxt - want); r, c = np.unravel_index(np.argmax(ee), ee.shape)
print("   worst row %d (deg At %d) got %.6e want %.6e ; dEd %.3e ; row terms max %.3e" % (r, degt[r], nxt[r, c], want[r, c], dEd[r, c],
      np.abs(At[r].data * dS[At[r].indices, c]).max()))
for rr in (2687, 13142):
    print("   row", rr, "nxt", nxt[rr, 10], "want", want[rr, 10], "dEd", dEd[rr, 10], "gE", gE[rr, 10], "dOut part", gE[rr, 10] - nxt[rr, 10])
# ---- replicate the oracle's backward in fp64, keep the intermediates
dt = np.float64
A_, At_ = A.astype(dt), At.astype(dt)
oE = E0.astype(dt); oW = [[w.astype(dt) for w in ws] for ws in W]
out, cache = O.ngcf_forward(A_, oE, [tuple(ws) for ws in oW], [m.astype(dt) for m in masks], 1 - drop)
iu, ii, ij = bu, U + bp, U + bn
eu, ei, ej = out[iu], out[ii], out[ij]
x = np.sum(eu * ei, 1) - np.sum(eu * ej, 1)
lb, g = O.bpr_terms(x)
dOut = np.zeros_like(out)
np.add.at(dOut, iu, g[:, None] * (ei - ej)); np.add.at(dOut, ii, g[:, None] * eu); np.add.at(dOut, ij, -g[:, None] * eu)
keep = 0.9
dEgo = np.zeros((U + I, 16))
inter = {}
for k in (1, 0):
    ego, S, T1, T2, Bi, Zd, ss, inv, norm, mask = cache[k]
    Wg, bg, Wb, bb = oW[k]
    dNorm = dOut[:, 16 * (k + 1):16 * (k + 2)]
    dot = np.sum(dNorm * norm, 1, keepdims=True)
    dZd = np.where(ss > 1e-12, (dNorm - norm * dot) * inv, dNorm * inv) + dEgo
    dZ = dZd * mask / keep
    dT1 = dZ * np.where(T1 > 0, 1.0, 0.2); dT2 = dZ * np.where(T2 > 0, 1.0, 0.2)
    dBi = dT2 @ Wb.T
    dS = dT1 @ Wg.T + dBi * ego
    dEd = dBi * S
    dEgo = dEd + At_ @ dS
    inter[k] = (dS, dEd, dEgo, dT1, dT2)
for name, got_, want_ in (("dEgo_1", eng.dEgo[1], inter[1][2]), ("dS_0", eng.dS, inter[0][0]), ("dEd_0", eng.dEd, inter[0][1]),
                          ("dT1_0", eng.dT1, inter[0][3]), ("dT2_0", eng.dT2, inter[0][4]), ("dEgo_0", eng.dEgo[0], inter[0][2])):
    a = got_.cpu().numpy(); e_ = np.abs(a - want_)
    r, c = np.unravel_index(np.argmax(e_), e_.shape)
    print("%-7s max abs err %.3e at row %d (|want| there %.3e, global max %.3e); row 2687 col 10: got %.6e want %.6e" % (
        name, e_.max(), r, abs(want_[r, c]), np.abs(want_).max(), a[2687, 10], want_[2687, 10]))
