"""GPU experiment: do 128-byte gathers confined to one half of 256-byte rows reach all L2 channels?"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd._lib import lib
import _explib
explib = _explib.load()

fn = explib.nrhip_exp_halfline
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
fn.restype = C.c_int
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bench(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


n = 1_620_000
out = torch.zeros(1 << 20, device="cuda")
for rows in (4096, 8192, 16384, 35000, 70839, 283356):
    T = torch.randn(rows, 64, device="cuda")
    ids = torch.randint(0, rows, (n,), device="cuda", dtype=torch.int32)
    res = []
    for mode in (0, 1, 2):
        res.append(bench(lambda: fn(C.c_void_p(ids.data_ptr()), n, 512, C.c_void_p(T.data_ptr()), mode, C.c_void_p(out.data_ptr()), st)))
    print("table %6d rows (%5.1f MB; touched half: %5.1f MB): half0 %.1f us  parity-mixed %.1f us  half1 %.1f us" % (
        rows, rows * 256 / 1e6, rows * 128 / 1e6, *res), flush=True)
