"""GPU experiment: score GEMM time per 16384-user batch at the gowalla shape (d=64)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E
rng = np.random.RandomState(0)
U, I, d, B = 29858, 40981, 64, 16384
P = torch.from_numpy((rng.randn(U, d) * 0.1).astype(np.float32)).cuda()
Q = torch.from_numpy((rng.randn(I, d) * 0.1).astype(np.float32)).cuda()
g = E.ScoreGemm(Q, B); S = g.new_score_buffer()
users = torch.arange(B, dtype=torch.int32, device="cuda")
for _ in range(3): g(P, users, out=S)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): g(P, users, out=S)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print("score GEMM: %.3f ms per %d-user batch  %.1f TFLOP/s  S write %.2f TB/s" % (
    ms, B, 2.0 * B * I * d / ms / 1e9, B * 41024 * 4 / ms / 1e9))
