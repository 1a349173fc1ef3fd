"""GPU micro-benchmark: random 256-byte row gathers from a [N][64] fp32 table — rows per load
instruction (1 / 2 / 4 via dword / dwordx2 / dwordx4), gathers in flight, list length per wave,
table size (L2-resident vs not)."""
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
n = 1_620_256
rng = np.random.RandomState(0)
print("table_rows table_MB vec in_flight per_wave : us  TB/s")
for N in (8000, 35000, 70839, 300000):
    T = torch.randn(N, 64, device="cuda")
    ids = torch.from_numpy(rng.randint(0, N, n).astype(np.int32)).cuda()
    out = torch.empty((n // 32 + 8) * 64, device="cuda")
    for vec, g in ((1, 16), (1, 8), (2, 8), (2, 16), (4, 4), (4, 8), (4, 16)):
        for per_wave in (64, 256, 1024):
            def run():
                rc = fn(ids.data_ptr(), n, per_wave, T.data_ptr(), vec, g, out.data_ptr(), st)
                assert rc == 0, lib.nrhip_last_error()
            for _ in range(3): run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): run()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 30 * 1e3
            print("%7d %6.1f  vec=%d G=%2d per_wave=%4d : %6.1f us  %5.2f TB/s"
                  % (N, N * 256 / 1e6, vec, g, per_wave, us, n * 256 / us / 1e6), flush=True)
