"""GPU micro-benchmark for the config-4 regime: random 256-byte row gathers at table sizes
around the 256 MB Infinity Cache — does a window that is Infinity-Cache-resident gather faster than one
that is HBM-resident?  If the L2-miss path (fabric) is the bound in both, column windows cannot pay for
the partial-row streams they add."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd._lib import lib
import _explib
explib = _explib.load()
fn = explib.nrhip_exp_gather
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
fn.restype = C.c_int
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n = 32_000_000
print("table_MB distinct_rows gathers : us  TB/s (256-B rows; ids uniform over the table)")
g = torch.Generator(device="cuda"); g.manual_seed(0)
for mb in (16, 64, 128, 192, 256, 384, 512, 1024, 2048, 5120):
    N = mb * 1_000_000 // 256
    T = torch.empty(N, 64, device="cuda").normal_()
    ids = torch.randint(0, N, (n,), generator=g, device="cuda", dtype=torch.int32)
    out = torch.empty((n // 32 + 8) * 64, device="cuda")
    for vec, gi, per_wave in ((4, 8, 256), (4, 16, 1024)):
        def run():
            rc = fn(ids.data_ptr(), n, per_wave, T.data_ptr(), vec, gi, out.data_ptr(), st)
            assert rc == 0, lib.nrhip_last_error()
        for _ in range(2): run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 5 * 1e3
        print("%6d MB %9d rows  vec=%d G=%2d per_wave=%4d : %9.1f us  %5.2f TB/s"
              % (mb, N, vec, gi, per_wave, us, n * 256 / us / 1e6), flush=True)
    del T, ids, out
