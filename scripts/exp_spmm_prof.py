"""GPU experiment: per-workgroup timeline of an SpMM kernel.  Needs a temporarily instrumented build:
a `__device__ unsigned long long g_prof[4096 * 24]` in spmm_blocked.hip, `wall_clock64()` stamps stored by thread 0
(and lane 0 of every wave) at the phase boundaries of the kernel under study, and an
`extern "C" int nrhip_exp_spmm_prof(void* host_out)` that copies the array out (hipMemcpyFromSymbol).  The results
of the round are in profiles/r01_exp_spmm_timeline.txt, r01_exp_halfrow_two_pass.txt, r01_exp_masked_hops.txt."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth, graph
from neurec_amd._lib import lib
if not hasattr(lib, "nrhip_exp_spmm_prof"):
    sys.exit("exp_spmm_prof.py needs an instrumented build of spmm_blocked.hip (see the docstring)")

tr, te = synth.interactions("gowalla")
coo = tr.tocoo(); U, I = tr.shape
A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
N, d = U + I, 64
X = torch.randn(N, d, device="cuda")
csr = E.SpmmCSR.from_scipy(A, split_row=U)
Y = torch.empty_like(X)
flag = torch.zeros(N, dtype=torch.uint8, device="cuda")
rs = np.random.RandomState(7)                         # a batch as the sampler draws it: 1024 interactions + negatives
pick = rs.randint(0, coo.nnz, 1024)
flag[torch.from_numpy(np.concatenate([coo.row[pick], U + coo.col[pick], U + rs.randint(0, I, 1024)]).astype(np.int64)).cuda()] = 1
Xs = X * flag[:, None].float()
buf = np.zeros(4096 * 24, dtype=np.uint64)
us = lambda x: x * 10.0 / 1e3
for name, Xi, kw in (("row-mask", X, {"y_row_wanted": flag}), ("col-mask", Xs, {"x_row_nonzero": flag, "addend": Xs})):
    for _ in range(5):
        csr.matmul(Xi, out=Y, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); csr.matmul(Xi, out=Y, **kw); b.record(); torch.cuda.synchronize()
    lib.nrhip_exp_spmm_prof(C.c_void_p(buf.ctypes.data))
    q = buf.reshape(4096, 24)[:256].astype(np.int64)
    t0 = q[:, 0].min()
    if name == "row-mask":
        end, marks = 2, [("scan+compact", 0, 1)]
    else:
        end, marks = 3, [("stage", 0, 1), ("compact(wave0)", 1, 2)]
    print("== %s: event time %.1f us; start skew %.2f us; last end %.2f us" % (name, a.elapsed_time(b) * 1e3, us(q[:, 0].max() - t0), us(q[:, end].max() - t0)))
    for nm, i, j in marks:
        v = us(q[:, j] - q[:, i]); print("  %-14s min %.2f med %.2f p90 %.2f max %.2f us" % (nm, v.min(), np.median(v), np.percentile(v, 90), v.max()))
    w = us(q[:, 8:24] - q[:, marks[-1][2]][:, None])
    print("  walk per wave: med-of-min %.2f med-of-max %.2f max %.2f us" % (np.median(w.min(1)), np.median(w.max(1)), w.max()))
    v = us(q[:, end] - q[:, 8:24].max(1)); print("  combine+tail   min %.2f med %.2f max %.2f us" % (v.min(), np.median(v), v.max()))
    v = us(q[:, end] - q[:, 0]); print("  total          min %.2f med %.2f p90 %.2f max %.2f us" % (v.min(), np.median(v), np.percentile(v, 90), v.max()))
    if name == "row-mask":
        order = np.argsort(q[:, end] - q[:, 0])[-4:]
        for wgi in order:
            print("   wg %3d wanted %3d total %.2f walk-max %.2f" % (wgi, q[wgi, 3], us(q[wgi, end] - q[wgi, 0]), w[wgi].max()))
