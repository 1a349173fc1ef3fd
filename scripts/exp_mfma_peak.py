"""GPU experiment: sustained rate of v_mfma_f32_32x32x2_f32 with nothing else in the loop (what the
evaluation scoring kernels should be compared with, rather than the data-sheet figure)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib
explib = _explib.load()
fn = explib.nrhip_exp_mfma_peak
fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(1 << 20, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for blocks, wps, iters in ((256, 1, 400), (2048, 1, 50), (256, 2, 200), (2106, 1, 37)):
    for _ in range(2):
        fn(blocks, wps, iters, C.c_void_p(out.data_ptr()), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn(blocks, wps, iters, C.c_void_p(out.data_ptr()), st)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    flops = blocks * 4 * wps * iters * 128 * 4096.0
    print("blocks %4d waves/SIMD %d iters %3d: %.3f ms  %.1f TFLOP/s" % (blocks, wps, iters, ms, flops / ms / 1e9))
