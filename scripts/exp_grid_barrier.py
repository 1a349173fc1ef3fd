"""What a device-wide barrier costs on the MI355X (a persistent BPR-MF epoch kernel would pay one per
step instead of a launch).  n_wg resident workgroups cross `iters` barriers; us per barrier = kernel time / iters."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib  # noqa: E402

lib = _explib.load()
fn = lib.nrhip_exp_grid_barrier
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
fn.restype = C.c_int
cnt = torch.zeros(16 * 9, dtype=torch.int32, device="cuda")
out = torch.zeros(4096, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(n_wg, iters, mode, work):
    assert fn(cnt.data_ptr(), n_wg, iters, mode, work, out.data_ptr(), st) == 0


def timed(n_wg, iters, mode, work):
    run(n_wg, 10, mode, work)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    run(n_wg, iters, mode, work)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3


print("device-wide barrier, us per barrier (kernel of 2,000 barriers minus the same kernel with 1): 256-thread workgroups")
for n_wg in (256, 469, 512, 1024):
    for mode, name in ((0, "one counter"), (1, "per-XCD + global")):
        base = timed(n_wg, 1, mode, 0)
        t = timed(n_wg, 2001, mode, 0)
        print("  %4d workgroups  %-18s %.2f us per barrier   (one-barrier kernel: %.1f us)"
              % (n_wg, name, (t - base) / 2000, base))
