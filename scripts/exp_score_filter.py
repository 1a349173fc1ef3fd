"""Level 1 of the pruned evaluation: the bf16 bounded filter against the fp32 MFMA loop (gowalla shape).
Prints both times, the largest |filter - fp32 chain| over all (user, tile) maxima relative to the row's bound, and how
many tile maxima differ at all."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurec_amd import engine as E

dev = torch.device("cuda:0")
torch.manual_seed(1)
U, I, d = 29858, 40981, int(os.environ.get("D", 64))
scale = float(os.environ.get("SCALE", 0.01))
P = (torch.randn(U, d, device=dev) * scale).contiguous()
Q = (torch.randn(I, d, device=dev) * scale).contiguous()
rows = 16384
users = torch.arange(rows, dtype=torch.int32, device=dev)
g = E.ScoreGemm(Q, rows)
f = E.ScoreFilter(Q, rows)


def hip_time(fn, n=10, w=3):
    for _ in range(w):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


class NoTrain:
    indptr = indices = None


def fp32():
    n_tiles = 2 * ((I + 63) // 64)
    mld = (n_tiles + 3) // 4 * 4
    out = torch.empty((rows, mld), dtype=torch.float32, device=dev)
    E.call("nrhip_score_tilemax", E._ptr(P), P.stride(0), E._ptr(users), rows, I, d, None, None, E._ptr(out),
           out.stride(0), E._ptr(g.ws), g.ws.numel(), E._stream())
    return out


M0 = fp32()
M1, eps = f.tile_maxima(P, users)
torch.cuda.synchronize()
n_tiles = 2 * ((I + 63) // 64)
A, B = M0[:, :n_tiles], M1[:, :n_tiles]
fin = torch.isfinite(A)
assert bool((torch.isfinite(B) == fin).all()), "pad tiles differ"
diff = torch.where(fin, (A - B).abs(), torch.zeros_like(A))
ratio = (diff / eps[:, None]).max().item()
print("d=%d  kappa %.3e  eps median %.3e  max|diff| %.3e  max diff/eps %.4f  maxima that differ: %.1f %%" % (
    d, f.kappa, eps.median().item(), diff.max().item(), ratio, 100.0 * (diff > 0).float().mean().item()))
t0 = hip_time(fp32)
t1 = hip_time(lambda: f.tile_maxima(P, users, out=M1, eps=eps))
tp = hip_time(lambda: f.prepare(Q))
flops = 2.0 * rows * I * d
print("fp32 loop %.3f ms (%.1f TFLOP/s)   bf16 filter %.3f ms (%.1f effective, %.0f issued bf16 TFLOP/s)   item split %.3f ms" % (
    t0, flops / t0 / 1e9, t1, flops / t1 / 1e9, 3 * flops / t1 / 1e9, tp))
