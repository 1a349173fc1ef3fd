"""The wide Mult-VAE's products through nrhip_gemm_f32 with and without transposed copies (HIP events):
k-major operands prepared by nrhip_transpose2d vs k-minor operands read as they lie."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd._lib import call
from neurec_amd.engine import _ptr, _stream

B, h, I = 512, 600, 40981
r = lambda *s: torch.randn(*s, device="cuda")
ws = torch.empty(64 * B * h * 4, dtype=torch.uint8, device="cuda")


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


def gemm(A, lda, ak, Bm, ldb, bk, M, N, K, C, ldc, sp=1):
    call("nrhip_gemm_f32", _ptr(A), lda, ak, _ptr(Bm), ldb, bk, M, N, K, _ptr(C), ldc, 0, None, -1, sp, _ptr(ws),
         ws.numel() if sp > 1 else 0, _stream())


def tr(src, ld, rows, cols, dst, ldd):
    call("nrhip_transpose2d", _ptr(src), ld, rows, cols, _ptr(dst), ldd, _stream())


fl = 2.0 * B * h * I
g, W, D = r(B, h), r(h, I), r(B, 41024)
gT, DT, WT = r(h, B), r(I, B), r(I, h)
S, dg = r(B, 41024), r(B, h)
t0 = timed(lambda: (tr(g, h, B, h, gT, B), gemm(gT, B, 0, W, I, 0, B, I, h, S, 41024)))
t1 = timed(lambda: gemm(g, h, 1, W, I, 0, B, I, h, S, 41024))
print("logits: transpose g + k-major x k-major %.1f us | g k-minor as it lies %.1f us (%.1f TFLOP/s)" % (t0, t1, fl / t1 / 1e6))
for sp in (16, 32, 64):
    t0 = timed(lambda: (tr(D, 41024, B, I, DT, B), tr(W, I, h, I, WT, h), gemm(DT, B, 0, WT, h, 0, B, h, I, dg, h, sp)))
    t1 = timed(lambda: gemm(D, 41024, 1, W, I, 1, B, h, I, dg, h, sp))
    print("dg splits %2d: 2 transposes + k-major x k-major %.1f us | both k-minor as they lie %.1f us (%.1f TFLOP/s)"
          % (sp, t0, t1, fl / t1 / 1e6))
X, Wm, Y = r(B, 600), r(600, 400), r(B, 400)
XT = r(600, B)
t0 = timed(lambda: (tr(X, 600, B, 600, XT, B), gemm(XT, B, 0, Wm, 400, 0, B, 400, 600, Y, 400)))
t1 = timed(lambda: gemm(X, 600, 1, Wm, 400, 0, B, 400, 600, Y, 400))
print("mid x W (512x600 . 600x400): with transpose %.1f us | x k-minor %.1f us" % (t0, t1))
dA, dX, dAT, WmT = r(B, 400), r(B, 600), r(400, B), r(400, 600)
t0 = timed(lambda: (tr(dA, 400, B, 400, dAT, B), tr(Wm, 400, 600, 400, WmT, 600), gemm(dAT, B, 0, WmT, 600, 0, B, 600, 400, dX, 600)))
t1 = timed(lambda: gemm(dA, 400, 1, Wm, 400, 1, B, 600, 400, dX, 600))
print("mid dA W^T (512x400 . 400x600): with 2 transposes %.1f us | both k-minor %.1f us" % (t0, t1))
N_ = 70839
S_, W64, T1 = r(N_, 64), r(64, 64), r(N_, 64)
ST = r(64, N_)
t0 = timed(lambda: (tr(S_, 64, N_, 64, ST, N_), gemm(ST, N_, 0, W64, 64, 0, N_, 64, 64, T1, 64)))
t1 = timed(lambda: gemm(S_, 64, 1, W64, 64, 0, N_, 64, 64, T1, 64))
print("ngcf S W (70839x64 . 64x64): with transpose %.1f us | S k-minor %.1f us" % (t0, t1))
