"""Row-masked hop (last forward hop of a step) under different wanted sets, old staged kernel
(NEUREC_SPMM_WANTED_WAVE=0) vs the wave-cooperative one (=1): where does the time go?"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth, graph
from neurec_amd._lib import lib

p, sz = C.c_void_p, C.c_size_t


def bench(fn, reps=100):
    for _ in range(5): fn()
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
deg = np.diff(A.indptr)
X = torch.randn(N, d, device="cuda"); Ssum = torch.randn_like(X)
indptr = np.ascontiguousarray(A.indptr.astype(np.int64)); indices = np.ascontiguousarray(A.indices.astype(np.int32))
ind_d = torch.from_numpy(indices).cuda(); val_d = torch.from_numpy(A.data.astype(np.float32)).cuda()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
rs = np.random.RandomState(7)
pick = rs.randint(0, coo.nnz, 1024)
batch_rows = np.unique(np.concatenate([coo.row[pick], U + coo.col[pick], U + rs.randint(0, I, 1024)]))
masks = {"empty": np.zeros(0, np.int64),
         "2.9k short rows (<=64)": rs.choice(np.flatnonzero(deg <= 64), 2900, replace=False),
         "50 biggest hubs": np.argsort(-deg)[:50],
         "batch (%d rows, %.0f%% of nnz)" % (len(batch_rows), 100.0 * deg[batch_rows].sum() / A.nnz): batch_rows,
         "all rows": np.arange(N)}
for mode in ("0", "1"):
    os.environ["NEUREC_SPMM_WANTED_WAVE"] = mode
    nb = sz(0); lib.nrhip_spmm_blocked_plan_bytes(N, A.nnz, d, C.byref(nb))
    buf = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
    plan = p()
    rc = lib.nrhip_spmm_blocked_plan_create(indptr.ctypes.data_as(p), indices.ctypes.data_as(p), N, U, d, 0, 0, 0, 0, 0, 0,
                                            ptr(buf), buf.numel(), st, C.byref(plan))
    assert rc == 0, lib.nrhip_last_error()
    for name, rows in masks.items():
        flag = torch.zeros(N, dtype=torch.uint8, device="cuda")
        flag[torch.from_numpy(np.asarray(rows, np.int64)).cuda()] = 1
        So = torch.zeros_like(X)
        fn = lambda: lib.nrhip_spmm_blocked(plan, ptr(ind_d), ptr(val_d), ptr(X), ptr(None), ptr(None), ptr(Ssum), ptr(So), ptr(None), ptr(flag), st)
        print("WAVE=%s  %-34s %.1f us" % (mode, name, bench(fn)), flush=True)
