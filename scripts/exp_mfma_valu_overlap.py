"""Do fp32 MFMAs of one wave and VALU work of another wave on the same SIMD overlap on the MI355X?
256 workgroups x 8 waves (2 per SIMD): waves 0-3 role A, waves 4-7 role B (csrc/experiments)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib  # noqa: E402

lib = _explib.load()
fn = lib.nrhip_exp_overlap
fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
fn.restype = C.c_int
out = torch.zeros(256 * 512, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
NAMES = {0: "idle", 1: "mfma x16", 2: "fma x256", 3: "exp x64", 4: "bf16mfma x16", 5: "bf16mfma x16 + 64 fma (one wave)", 6: "fma x64 (4 chains)"}


def timed(a, b, iters=2000):
    fn(256, a, b, 50, out.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(256, a, b, iters, out.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters          # us per iteration


print("us per iteration (one iteration = 16 dependent MFMA 32x32x2 f32 | 256 dependent fma | 64 dependent exp+fma)")
for a, b in ((1, 0), (0, 2), (0, 3), (1, 1), (2, 2), (3, 3), (1, 2), (1, 3), (2, 3)):
    print("  waves 0-3: %-9s waves 4-7: %-9s  %.3f us" % (NAMES[a], NAMES[b], timed(a, b)))

print("bf16 MFMA 32x32x16 (16 per iteration over four accumulators) and VALU")
for a, b in ((4, 0), (0, 6), (5, 0), (4, 6), (4, 4), (5, 5), (4, 2)):
    print("  waves 0-3: %-34s waves 4-7: %-18s  %.3f us" % (NAMES[a], NAMES[b], timed(a, b)))
