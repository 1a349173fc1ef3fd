"""time(K) of the logits-shaped product (M = 512, N = 40,981): fixed per-launch / per-block cost vs the k-loop's rate"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd._lib import call
from neurec_amd.engine import _ptr, _stream
B, I = 512, 40981
ldc = int(sys.argv[1]) if len(sys.argv) > 1 else I
r = lambda *s: torch.randn(*s, device="cuda")
S = r(B, ldc)
for K in (16, 150, 300, 600, 1200, 2400):
    A, W = r(K, B), r(K, I)
    def fn():
        call("nrhip_gemm_kmajor", _ptr(A), B, _ptr(W), I, B, I, K, _ptr(S), ldc, 0, None, -1, 1, None, 0, _stream())
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 10 * 1e3
    print("K=%5d ldc=%d: %.1f us, %.1f TFLOP/s" % (K, ldc, t, 2.0 * B * I * K / t / 1e6))
