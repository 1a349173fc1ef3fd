"""GPU experiment: the dedicated masked-hop kernels (wanted rows / column-masked) against the general
masked lane-group kernel: bit-equality of every variant and time per pass."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth, graph
from neurec_amd._lib import lib

p, sz = C.c_void_p, C.c_size_t


def bench(fn, reps=50):
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
X = torch.randn(N, d, device="cuda"); Ssum = torch.randn_like(X); Add = torch.randn_like(X)
indptr = np.ascontiguousarray(A.indptr.astype(np.int64)); indices = np.ascontiguousarray(A.indices.astype(np.int32))
ind_d = torch.from_numpy(indices).cuda(); val_d = torch.from_numpy(A.data.astype(np.float32)).cuda()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
nb = sz(0); lib.nrhip_spmm_blocked_plan_bytes(N, A.nnz, d, C.byref(nb))
flag = torch.zeros(N, dtype=torch.uint8, device="cuda")
rs = np.random.RandomState(7)                         # a batch as the sampler draws it: 1024 interactions + negatives
pick = rs.randint(0, coo.nnz, 1024)
flag[torch.from_numpy(np.concatenate([coo.row[pick], U + coo.col[pick], U + rs.randint(0, I, 1024)]).astype(np.int64)).cuda()] = 1
Xs = X * flag[:, None].float()
plans = {}
for mode in ("0", "1"):
    os.environ["NEUREC_SPMM_MASKED_FAST"] = mode
    buf = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
    plan = p()
    rc = lib.nrhip_spmm_blocked_plan_create(indptr.ctypes.data_as(p), indices.ctypes.data_as(p), N, U, d, 0, 0, 0, 0, 0, 0,
                                            ptr(buf), buf.numel(), st, C.byref(plan))
    assert rc == 0, lib.nrhip_last_error()
    plans[mode] = (plan, buf)


def run(mode, Xi, Yo, ad=None, si=None, so=None, cm=None, rm=None):
    rc = lib.nrhip_spmm_blocked(plans[mode][0], ptr(ind_d), ptr(val_d), ptr(Xi), ptr(Yo), ptr(ad), ptr(si), ptr(so), ptr(cm), ptr(rm), st)
    assert rc == 0, lib.nrhip_last_error()


cases = {"plain": dict(Xi=X), "col-mask": dict(Xi=Xs, cm=flag), "col-mask+addend=X": dict(Xi=Xs, ad=Xs, cm=flag),
         "col-mask+addend+sum": dict(Xi=Xs, ad=Add, si=Ssum, cm=flag), "row-mask": dict(Xi=X, rm=flag),
         "row-mask+sum(noY)": dict(Xi=X, si=Ssum, rm=flag, noY=True), "row-mask+addend": dict(Xi=X, ad=Add, rm=flag)}
for name, kw in cases.items():
    out, t = {}, {}
    for mode in ("0", "1"):
        kw2 = {k: v for k, v in kw.items() if k != "noY"}
        Y = None if kw.get("noY") else torch.zeros_like(X); So = torch.zeros_like(X) if "si" in kw else None
        run(mode, Yo=Y, so=So, **kw2); torch.cuda.synchronize()
        out[mode] = (Y if Y is not None else So, So)
        t[mode] = bench(lambda: run(mode, Yo=Y, so=So, **kw2))
    same = torch.equal(out["0"][0], out["1"][0]) and (out["0"][1] is None or torch.equal(out["0"][1], out["1"][1]))
    print("%-20s general %.1f us  dedicated %.1f us  bit-identical=%s" % (name, t["0"], t["1"], same), flush=True)
