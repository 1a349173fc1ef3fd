"""GPU experiment: per-workgroup timeline of spmm_blocked_kernel (lane-group SpMM) at a given width.  Needs the
instrumented build:   NEUREC_HIPCC_EXTRA=-DNR_BLK_TIMELINE python -m neurec_amd.build --force
(rebuild without the variable afterwards).  Stamps: 0 start, 1 LDS zeroed, 8..23 each wave's end of walk,
2 start of the hub combine, 3 before the epilogue, 4 end."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth, graph
from neurec_amd._lib import lib
if not hasattr(lib, "nrhip_exp_blk_timeline"):
    sys.exit("needs the instrumented build (see the docstring)")
tr, _ = synth.interactions("gowalla")
coo = tr.tocoo(); U, I = tr.shape
A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
buf = np.zeros(4096 * 24, dtype=np.uint64)
us = lambda x: x * 10.0 / 1e3                       # wall_clock64: 100 MHz
for d in (16, 32, 64):
    X = torch.randn(U + I, d, device="cuda"); Y = torch.empty_like(X)
    csr = E.SpmmCSR.from_scipy(A, split_row=U)
    for _ in range(5): csr.matmul(X, out=Y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); csr.matmul(X, out=Y); b.record(); torch.cuda.synchronize()
    lib.nrhip_exp_blk_timeline(C.c_void_p(buf.ctypes.data))
    q = buf.reshape(4096, 24)[:256].astype(np.int64)
    t0 = q[:, 0].min()
    print("== d=%d: event %.1f us; first start -> last end %.2f us; start skew %.2f us" % (d, a.elapsed_time(b) * 1e3, us(q[:, 4].max() - t0), us(q[:, 0].max() - t0)))
    for nm, i, j in (("zero LDS", 0, 1), ("epilogue", 3, 4), ("total", 0, 4)):
        v = us(q[:, j] - q[:, i]); print("  %-10s min %.2f med %.2f p90 %.2f max %.2f" % (nm, v.min(), np.median(v), np.percentile(v, 90), v.max()))
    w = us(q[:, 8:24] - q[:, 1][:, None])
    print("  walk per wave: min %.2f med-of-med %.2f med-of-max %.2f max %.2f" % (w.min(), np.median(np.median(w, 1)), np.median(w.max(1)), w.max()))
    v = us(q[:, 3] - q[:, 8:24].max(1)); print("  last wave done -> epilogue (barrier + hub combine): min %.2f med %.2f max %.2f" % (v.min(), np.median(v), v.max()))
