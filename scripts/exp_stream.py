"""GPU experiment: how fast can a [16384][41024] fp32 slab be read back from HBM?  flat sum vs
row-wise max (torch kernels) vs the select kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E
B, ld = 16384, 41024
S = torch.randn(B, ld, device="cuda")
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
gb = B * ld * 4 / 1e9
for name, fn in (("flat sum", lambda: S.sum()), ("row max", lambda: S.amax(dim=1)),
                 ("arg_topk 40", lambda: E.arg_topk(S, 40, cols=40981))):
    ms = t(fn)
    print("%-12s %.3f ms  %.2f TB/s" % (name, ms, gb / ms))

import ctypes as C
from neurec_amd._lib import lib
import _explib
explib = _explib.load()
fn = explib.nrhip_exp_rowmax
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
fn.restype = C.c_int
out = torch.empty(B * 4, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for wpr, u in ((1, 4), (1, 8), (1, 16), (2, 8), (4, 2), (4, 4), (4, 8)):
    def run():
        assert fn(S.data_ptr(), ld, B, 40981, wpr, u, out.data_ptr(), st) == 0
    ms = t(run)
    print("rowmax waves/row=%d in_flight=%2d : %.3f ms  %.2f TB/s" % (wpr, u, ms, gb / ms))
