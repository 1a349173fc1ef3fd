"""The three item-layer products of the wide Mult-VAE (B = 512, h = 600, I = 40,981) through nrhip_gemm_kmajor,
timed one by one with HIP events.  usage (through gpurun): python scripts/exp_gemm.py [splits of dg ...]"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd._lib import call
from neurec_amd.engine import _ptr, _stream

B, h, I = 512, 600, 40981
dev = "cuda"
r = lambda *s: torch.randn(*s, device=dev)
gT, W, g, D, DT, WT = r(h, B), r(h, I), r(B, h), r(B, I), r(I, B), r(I, h)
S, dW, dg = r(B, I), r(h, I), r(B, h)
ws = torch.empty(64 * B * h * 4, dtype=torch.uint8, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def gemm(A, lda, Bm, ldb, M, N, K, Cm, ldc, splits=1):
    call("nrhip_gemm_kmajor", _ptr(A), lda, _ptr(Bm), ldb, M, N, K, _ptr(Cm), ldc, 0, None, -1, splits, _ptr(ws),
         ws.numel() if splits > 1 else 0, _stream())


fl = 2.0 * B * h * I
t = timed(lambda: gemm(gT, B, W, I, B, I, h, S, I))
print("logits  M=%d N=%d K=%d: %.1f us, %.1f TFLOP/s" % (B, I, h, t, fl / t / 1e6))
t = timed(lambda: gemm(g, h, D, I, h, I, B, dW, I))
print("dW      M=%d N=%d K=%d: %.1f us, %.1f TFLOP/s" % (h, I, B, t, fl / t / 1e6))
for sp in [int(x) for x in sys.argv[1:]] or [16, 32, 38, 51]:
    t = timed(lambda: gemm(DT, B, WT, h, B, h, I, dg, h, sp))
    print("dg      M=%d N=%d K=%d splits %d: %.1f us, %.1f TFLOP/s" % (B, h, I, sp, t, fl / t / 1e6))
for (M, N, K, sp) in ((512, 400, 600, 1), (512, 400, 600, 4), (600, 400, 512, 4), (512, 600, 400, 1), (512, 600, 200, 1)):
    A_, B_, C_ = r(K, M), r(K, N), r(M, N)
    t = timed(lambda: gemm(A_, M, B_, N, M, N, K, C_, N, sp))
    print("mid     M=%d N=%d K=%d splits %d: %.1f us, %.1f TFLOP/s" % (M, N, K, sp, t, 2.0 * M * N * K / t / 1e6))
