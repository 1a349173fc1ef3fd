"""GPU micro-benchmark: column-blocked persistent gathers on the row-major [N][64] table (the
premise of a cache-blocked SpMM): 256 workgroups, class pinned to XCD halves, K column blocks."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd._lib import lib
import _explib
explib = _explib.load()

fn = explib.nrhip_exp_gather_blocked
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
fn.restype = C.c_int
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
U, I = 29858, 40981
N = U + I
total = 1_620_256
rng = np.random.RandomState(0)
T = torch.randn(N, 64, device="cuda")
out = torch.empty(512 * 16 * 64, device="cuda")
print("n_wg K vec G waves : us  TB/s")
for n_wg in (256, 512):
    for K in (1, 2, 4, 8):
        per_phase = total // n_wg // K
        ids = np.empty((n_wg, K, per_phase), np.int32)
        for b in range(n_wg):
            cls = (b & 7) >> 2
            lo, hi = (U, N) if cls == 0 else (0, U)
            edges = np.linspace(lo, hi, K + 1).astype(np.int64)
            for k in range(K):
                ids[b, k] = rng.randint(edges[k], edges[k + 1], per_phase)
        ids_d = torch.from_numpy(ids).cuda()
        for vec, g, waves in ((1, 16, 16), (1, 8, 16), (4, 4, 16), (4, 8, 16), (4, 2, 16), (4, 4, 8), (1, 16, 8)):
            def run():
                rc = fn(ids_d.data_ptr(), n_wg, K, per_phase, T.data_ptr(), vec, g, waves, out.data_ptr(), st)
                assert rc == 0, lib.nrhip_last_error()
            for _ in range(3): run()
            torch.cuda.synchronize()
            a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): run()
            b2.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b2) / 30 * 1e3
            print("wg=%d K=%d vec=%d G=%2d waves=%2d : %6.1f us  %5.2f TB/s"
                  % (n_wg, K, vec, g, waves, us, n_wg * K * per_phase * 256 / us / 1e6), flush=True)
