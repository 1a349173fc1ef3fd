"""GPU experiment: the lane-group SpMM at narrow rows (d = 16 / 32: NGCF's layers, LightGCN at small embed_size) —
segment length, waves per workgroup, workgroups.  The pass is latency-bound there (4.5 MB table, 13.6 MB of CSR):
a 4-lane group walks a sub-list alone, so the longest sub-list (seg_len non-zeros) is the critical path."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth
from neurec_amd._lib import call
from neurec_amd.engine import _ptr, _stream
from neurec_amd.graph import ngcf_adjacency, lightgcn_adjacency

tr, _ = synth.interactions("gowalla", seed=2018)
U, I = tr.shape
coo = tr.tocoo()
mats = {"ngcf-norm": ngcf_adjacency(tr, "norm"), "lightgcn-pre": lightgcn_adjacency(coo.row, coo.col, U, I, "pre")}


def timed(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, A in mats.items():
    for d in (16, 32):
        X = torch.randn(A.shape[0], d, device="cuda")
        ref = None
        for seg, waves, nwg in ((0, 0, 0), (32, 0, 0), (16, 0, 0), (32, 8, 0), (16, 8, 0), (32, 0, 512), (16, 0, 512),
                                (16, 8, 1024), (32, 8, 1024)):
            csr = E.SpmmCSR.from_scipy(A, split_row=U)
            nbytes = C.c_size_t(0)
            call("nrhip_spmm_blocked_plan_bytes", csr.n_rows, csr.nnz, d, C.byref(nbytes))
            buf = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device="cuda")
            plan = C.c_void_p(0)
            try:
                call("nrhip_spmm_blocked_plan_create", csr.h_indptr.ctypes.data_as(C.c_void_p),
                     csr.h_indices.ctypes.data_as(C.c_void_p), csr.n_rows, csr.split_row, d, 0, nwg, waves, seg, 0, 0,
                     _ptr(buf), buf.numel(), _stream(), C.byref(plan))
            except (NotImplementedError, ValueError) as e:
                print("%-13s d=%d seg=%2d waves=%2d wg=%4d : %s" % (name, d, seg, waves, nwg, str(e)[:80]))
                continue
            call("nrhip_spmm_plan_attach_blocked", csr.plan, plan, d)
            csr._blocked[d] = (plan, buf)
            csr.blocked = plan
            Y = torch.empty_like(X)
            us = timed(lambda: csr.matmul(X, out=Y))
            if ref is None:
                ref = Y.clone()
            err = float((Y - ref).abs().max())
            print("%-13s d=%d seg=%2d waves=%2d wg=%4d : %6.2f us  (max |diff| vs default %.1e)" % (name, d, seg, waves, nwg, us, err), flush=True)
