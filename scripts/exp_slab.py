"""GPU experiment: slab-major SpMM (csrc/experiments/gather_experiments.hip) against the row-major work-item kernel on the
gowalla-shaped graph: correctness (vs the row-major result) and time per pass, for slab widths
8/16/32/64 and hub segment lengths."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth, graph
from neurec_amd._lib import lib
import _explib
explib = _explib.load()


def bench(fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def plan(indptr, split_row, seg_len, sort=True):
    n = len(indptr) - 1
    lens = np.diff(indptr)
    ent = {"row": [], "begin": [], "len": [], "slot": []}
    multi = {"row": [], "first": [], "nseg": []}
    counts = []
    slot = 0
    for lo, hi in ((0, split_row), (split_row, n)):
        rows, begins, ls, slots = [], [], [], []
        for r in range(lo, hi):
            L = int(lens[r])
            if L > seg_len:
                ns = (L + seg_len - 1) // seg_len
                multi["row"].append(r); multi["first"].append(slot); multi["nseg"].append(ns)
                for s in range(ns):
                    rows.append(r); begins.append(int(indptr[r]) + s * seg_len)
                    ls.append(min(seg_len, L - s * seg_len)); slots.append(slot); slot += 1
            else:
                rows.append(r); begins.append(int(indptr[r])); ls.append(L); slots.append(-1)
        order = np.argsort(-np.asarray(ls), kind="stable") if sort else np.arange(len(ls))
        for k, v in (("row", rows), ("begin", begins), ("len", ls), ("slot", slots)):
            ent[k].append(np.asarray(v)[order])
        counts.append(len(rows))
    cat = lambda k, t: torch.from_numpy(np.concatenate(ent[k]).astype(t)).cuda()
    arr = lambda k, t: torch.from_numpy(np.asarray(multi[k] or [0]).astype(t)).cuda()
    return (cat("row", np.int32), cat("begin", np.int64), cat("len", np.int32), cat("slot", np.int32),
            counts[0], counts[1], arr("row", np.int32), arr("first", np.int32), arr("nseg", np.int32),
            len(multi["row"]), slot)


fn = explib.nrhip_spmm_slab
p, i32, i64 = C.c_void_p, C.c_int, C.c_int64
fn.argtypes = [p, p, p, p, i32, i32, p, p, p, i32, p, p, p, i64, i32, i32, i32, p, p, p, p, p, p]
fn.restype = C.c_int

tr, te = synth.interactions("gowalla")
coo = tr.tocoo(); U, I = tr.shape
A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
N, d = U + I, 64
X = torch.randn(N, d, device="cuda")
Ssum = torch.randn_like(X)
csr = E.SpmmCSR.from_scipy(A, split_row=U)
Yref = torch.empty_like(X); Sref = Ssum.clone()
csr.matmul(X, out=Yref, sum_in=Ssum, sum_out=Sref)
us = bench(lambda: csr.matmul(X, out=Yref))
print("row-major work-item kernel: %.1f us" % us, flush=True)
indices = torch.from_numpy(A.indices.astype(np.int32)).cuda()
vals = torch.from_numpy(A.data.astype(np.float32)).cuda()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

for sw, wpb in ((16, 4), (16, 8), (16, 2), (8, 4), (32, 4), (64, 4)):
    S = d // sw
    to_slab = lambda T: T.view(N, S, sw).permute(1, 0, 2).contiguous()
    from_slab = lambda T: T.view(S, N, sw).permute(1, 0, 2).reshape(N, d)
    Xs, Ss = to_slab(X), to_slab(Ssum)
    for seg in (64, 128, 256):
        for sort in (True, False):
            er, eb, el, es, na, nb, mr, mf, mn, nm, nslots = plan(A.indptr, U, seg, sort)
            Ys = torch.empty_like(Xs); So = torch.empty_like(Xs)
            part = torch.empty(max(nslots, 1) * d, device="cuda")
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
            def run(Yo=Ys, si=None, so=None):
                rc = fn(ptr(er), ptr(eb), ptr(el), ptr(es), na, nb, ptr(mr), ptr(mf), ptr(mn), nm,
                        ptr(indices), ptr(vals), ptr(Xs), N, d, sw, wpb, ptr(Yo), ptr(None), ptr(si),
                        ptr(so), ptr(part), st)
                assert rc == 0, lib.nrhip_last_error()
            run(Ys, Ss, So)
            torch.cuda.synchronize()
            err = (from_slab(Ys) - Yref).abs().max().item()
            err2 = (from_slab(So) - Sref).abs().max().item()
            exact = torch.equal(from_slab(Ys), Yref)
            us = bench(lambda: run(Ys))
            print("SW=%2d WPB=%d seg=%3d sort=%d entries=%d+%d hubs=%d : %.1f us  max|dY|=%.2e |dS|=%.2e exact=%s"
                  % (sw, wpb, seg, sort, na, nb, nm, us, err, err2, exact), flush=True)
