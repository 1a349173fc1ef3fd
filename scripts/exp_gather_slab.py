"""GPU micro-benchmark: the gathers of one gowalla-shaped SpMM pass on the slab-major layout
T[4][N][16] with (slab, class) pairs pinned to XCDs — is the per-XCD slice L2-resident, and what
does a 64-byte-piece gather cost per load shape?"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd._lib import lib
import _explib
explib = _explib.load()

fn = explib.nrhip_exp_gather_slab
fn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
fn.restype = C.c_int
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
U, I = 29858, 40981
N = U + I
half = 810_128
rng = np.random.RandomState(0)
ids = np.concatenate([rng.randint(U, N, half), rng.randint(0, U, half)]).astype(np.int32)
ids_d = torch.from_numpy(ids).cuda()
T = torch.randn(4, N, 16, device="cuda")
out = torch.empty((half // 16 + 64) * 64 * 8, device="cuda")
print("vec in_flight per_wave : us   (4 slabs x 1.62M pieces of 64 B = 415 MB)")
for vec, g in ((1, 16), (1, 8), (2, 8), (4, 2), (4, 4), (4, 8)):
    for per_wave in (64, 256, 1024):
        def run():
            rc = fn(ids_d.data_ptr(), half, per_wave, T.data_ptr(), N, vec, g, out.data_ptr(), st)
            assert rc == 0, lib.nrhip_last_error()
        for _ in range(3): run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): run()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 30 * 1e3
        print("vec=%d G=%2d per_wave=%4d : %6.1f us  %5.2f TB/s" % (vec, g, per_wave, us, 2 * half * 256 / us / 1e6), flush=True)
