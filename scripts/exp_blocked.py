"""GPU experiment: cache-blocked persistent SpMM (spmm_blocked.hip) vs the work-item kernel on the
gowalla-shaped graph: bit-equality on non-split rows, masks, and time per pass by block size."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth, graph
from neurec_amd._lib import lib

p, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
lib.nrhip_spmm_blocked_plan_bytes.argtypes = [i64, i64, C.POINTER(sz)]
lib.nrhip_spmm_blocked_plan_create.argtypes = [p, p, i64, i64, i32, i64, i32, i32, i32, i32, i32, p, sz, p, C.POINTER(p)]
lib.nrhip_spmm_blocked_tune.argtypes = [i32]
lib.nrhip_spmm_blocked_plan_info.argtypes = [p, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)]
lib.nrhip_spmm_blocked.argtypes = [p] * 11
for f in ("nrhip_spmm_blocked_plan_bytes", "nrhip_spmm_blocked_plan_create", "nrhip_spmm_blocked_plan_info", "nrhip_spmm_blocked"):
    getattr(lib, f).restype = C.c_int


def bench(fn, reps=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tr, te = synth.interactions("gowalla")
coo = tr.tocoo(); U, I = tr.shape
A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
N, d = U + I, 64
X = torch.randn(N, d, device="cuda"); Ssum = torch.randn_like(X); Add = torch.randn_like(X)
csr = E.SpmmCSR.from_scipy(A, split_row=U)
Yref = torch.empty_like(X); Sref = torch.empty_like(X)
csr.matmul(X, out=Yref, addend=Add, sum_in=Ssum, sum_out=Sref)
print("work-item kernel: %.1f us" % bench(lambda: csr.matmul(X, out=Yref, addend=Add, sum_in=Ssum, sum_out=Sref)), flush=True)
lens = np.diff(A.indptr)
short = torch.from_numpy(lens <= 32).cuda()          # rows that can never be split by either kernel
indptr = np.ascontiguousarray(A.indptr.astype(np.int64)); indices = np.ascontiguousarray(A.indices.astype(np.int32))
ind_d = torch.from_numpy(indices).cuda(); val_d = torch.from_numpy(A.data.astype(np.float32)).cuda()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
nb = sz(0); lib.nrhip_spmm_blocked_plan_bytes(N, A.nnz, C.byref(nb))
# masks as in the LightGCN step: ~3*1024 flagged rows
flag = torch.zeros(N, dtype=torch.uint8, device="cuda")
flag[torch.randint(0, N, (3072,), device="cuda")] = 1
Xs = X * flag[:, None].float()
Ym_ref = torch.zeros_like(X); csr.matmul(Xs, out=Ym_ref, x_row_nonzero=flag)
Yr_ref = torch.zeros_like(X); csr.matmul(X, out=Yr_ref, y_row_wanted=flag)
print("work-item masked: col %.1f us, row %.1f us" % (
    bench(lambda: csr.matmul(Xs, out=Ym_ref, x_row_nonzero=flag)),
    bench(lambda: csr.matmul(X, out=Yr_ref, y_row_wanted=flag))), flush=True)
for split, bb, waves, seg, gif in [(U, 1 << 20, 16, 64, 8), (U, 6144, 16, 32, 8), (U, 6144, 16, 16, 8),
                                  (U, 4096, 16, 32, 8), (U, 4096, 16, 16, 8), (U, 4096, 16, 16, 4),
                                  (U, 3072, 16, 32, 8), (U, 3072, 16, 16, 8), (U, 2560, 16, 16, 8),
                                  (U, 1 << 20, 16, 32, 8), (U, 1 << 20, 16, 16, 8)]:
        lib.nrhip_spmm_blocked_tune(gif)
        buf = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
        plan = p()
        rc = lib.nrhip_spmm_blocked_plan_create(indptr.ctypes.data, indices.ctypes.data, N, split, d, bb * 1024, 0,
                                                waves, seg, 0, 0, ptr(buf), buf.numel(), st, C.byref(plan))
        if rc != 0:
            print("split=%d block=%dKB waves=%d seg=%d: plan failed: %s" % (split, bb, waves, seg, lib.nrhip_last_error().decode())); continue
        nwg, nph, nent, ncmb = i32(0), i32(0), i64(0), i64(0)
        lib.nrhip_spmm_blocked_plan_info(plan, C.byref(nwg), C.byref(nph), C.byref(nent), C.byref(ncmb))
        Y = torch.empty_like(X); So = torch.empty_like(X)
        def run(Xi=X, Yo=Y, ad=Add, si=Ssum, so=So, cm=None, rm=None):
            rc = lib.nrhip_spmm_blocked(plan, ptr(ind_d), ptr(val_d), ptr(Xi), ptr(Yo), ptr(ad), ptr(si), ptr(so), ptr(cm), ptr(rm), st)
            assert rc == 0, lib.nrhip_last_error()
        run(); torch.cuda.synchronize()
        err = (Y - Yref).abs().max().item(); errs = (So - Sref).abs().max().item()
        exact_short = torch.equal(Y[short], Yref[short])
        Ym = torch.zeros_like(X); run(Xs, Ym, None, None, None, flag, None)
        Yr = torch.zeros_like(X); run(X, Yr, None, None, None, None, flag)
        torch.cuda.synchronize()
        em = (Ym - Ym_ref).abs().max().item(); er = (Yr - Yr_ref).abs().max().item()
        t_full = bench(run)
        t_sum = bench(lambda: run(X, Y, None, Ssum, So))
        t_add = bench(lambda: run(X, Y, Add, None, None))
        t_plain = bench(lambda: run(X, Y, None, None, None))
        t_col = bench(lambda: run(Xs, Ym, None, None, None, flag, None))
        t_row = bench(lambda: run(X, Yr, None, None, None, None, flag))
        print("split=%5d block=%7dKB waves=%2d seg=%3d G=%d wg=%d phases=%2d entries=%d splits=%d : add+sum %.1f sum %.1f add %.1f plain %.1f col-mask %.1f row-mask %.1f us | "
              "max|dY|=%.1e |dS|=%.1e short-exact=%s mask err %.1e %.1e"
              % (split, bb, waves, seg, gif, nwg.value, nph.value, nent.value, ncmb.value, t_full, t_sum, t_add, t_plain, t_col, t_row, err, errs, exact_short, em, er), flush=True)
        lib.nrhip_spmm_blocked_plan_destroy(plan)
