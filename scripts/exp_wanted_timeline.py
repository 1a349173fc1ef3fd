"""Per-workgroup phase timeline of spmm_wanted_wave_kernel (needs a library built with
-DNR_WW_TIMELINE: scripts/exp_wanted_timeline.sh).  Phases: bitmap | scan | walk | chunk combine."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NEUREC_SPMM_WANTED_WAVE"] = "1"
from neurec_amd import engine as E, synth, graph
from neurec_amd._lib import lib

p, sz = C.c_void_p, C.c_size_t
tr, te = synth.interactions("gowalla")
coo = tr.tocoo(); U, I = tr.shape
A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
N, d = U + I, 64
deg = np.diff(A.indptr)
X = torch.randn(N, d, device="cuda"); Ssum = torch.randn_like(X)
indptr = np.ascontiguousarray(A.indptr.astype(np.int64)); indices = np.ascontiguousarray(A.indices.astype(np.int32))
ind_d = torch.from_numpy(indices).cuda(); val_d = torch.from_numpy(A.data.astype(np.float32)).cuda()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
rs = np.random.RandomState(7)
pick = rs.randint(0, coo.nnz, 1024)
batch_rows = np.unique(np.concatenate([coo.row[pick], U + coo.col[pick], U + rs.randint(0, I, 1024)]))
nb = sz(0); lib.nrhip_spmm_blocked_plan_bytes(N, A.nnz, d, C.byref(nb))
buf = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
plan = p()
assert lib.nrhip_spmm_blocked_plan_create(indptr.ctypes.data_as(p), indices.ctypes.data_as(p), N, U, d, 0, 0, 0, 0, 0, 0,
                                          ptr(buf), buf.numel(), st, C.byref(plan)) == 0
for name, rows in (("batch", batch_rows), ("50 hubs", np.argsort(-deg)[:50]),
                   ("short", rs.choice(np.flatnonzero(deg <= 64), 2900, replace=False))):
    flag = torch.zeros(N, dtype=torch.uint8, device="cuda")
    flag[torch.from_numpy(np.asarray(rows, np.int64)).cuda()] = 1
    So = torch.zeros_like(X)
    for _ in range(3):
        lib.nrhip_spmm_blocked(plan, ptr(ind_d), ptr(val_d), ptr(X), ptr(None), ptr(None), ptr(Ssum), ptr(So), ptr(None), ptr(flag), st)
    out = (C.c_ulonglong * (256 * 8))()
    lib.nrhip_ww_timeline.argtypes = [C.c_void_p]
    assert lib.nrhip_ww_timeline(out) == 0
    t = np.frombuffer(out, dtype=np.uint64).reshape(256, 8).astype(np.int64)
    t0 = t[:, 0].min()
    ph = np.diff(t[:, :5], axis=1) * 0.01                      # 100 MHz ticks -> us
    print("%-8s start spread %.1f us | bitmap %.1f/%.1f  scan %.1f/%.1f  walk %.1f/%.1f  combine %.1f/%.1f (mean/max us) | "
          "end max %.1f us | wanted entries/wg mean %.1f max %d | chunks/wg max %d"
          % (name, (t[:, 0].max() - t0) * 0.01, ph[:, 0].mean(), ph[:, 0].max(), ph[:, 1].mean(), ph[:, 1].max(),
             ph[:, 2].mean(), ph[:, 2].max(), ph[:, 3].mean(), ph[:, 3].max(), (t[:, 4].max() - t0) * 0.01,
             t[:, 6].mean(), t[:, 6].max(), t[:, 7].max()))
