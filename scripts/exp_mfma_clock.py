import ctypes as C, os, sys, torch
sys.path.insert(0, "scripts")
import _explib
lib = _explib.load()
fn = lib.nrhip_exp_overlap
fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(256 * 512, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timed(blocks, a, b, iters=4000):
    fn(blocks, a, b, 50, out.data_ptr(), st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(blocks, a, b, iters, out.data_ptr(), st); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for blocks in (8, 32, 64, 128, 256):
    t = timed(blocks, 4, 0)
    t7 = timed(blocks, 7, 0)
    print("blocks %3d: 32 bf16 MFMA 16x16x32 per wave (same flops): %.3f us -> %.1f clk each" % (blocks, t7, t7 * 2400 / 32))
    print("blocks %3d: 16 bf16 MFMA 32x32x16 per wave: %.3f us -> %.1f clk each at 2.4 GHz; fp32 mfma x16: %.3f us" % (blocks, t, t * 2400 / 16, timed(blocks, 1, 0)))

print("the same with operands of random sign / mantissa bits (data-dependent power: does the cadence hold?)")
for blocks in (8, 256):
    t4, t7 = timed(blocks, 4, 0, 4001), timed(blocks, 7, 0, 4001)
    print("blocks %3d: 32x32x16 %.1f clk each, 16x16x32 %.1f clk each (at 2.4 GHz)" % (blocks, t4 * 2400 / 16, t7 * 2400 / 32))

print("int8 MFMAs (twice the MACs per instruction of the bf16 forms), random operand bits")
for blocks in (8, 256):
    t8, t9 = timed(blocks, 8, 0, 4001), timed(blocks, 9, 0, 4001)
    print("blocks %3d: i32_32x32x32_i8 %.1f clk each, i32_16x16x64_i8 %.1f clk each (at 2.4 GHz)" % (blocks, t8 * 2400 / 16, t9 * 2400 / 32))
