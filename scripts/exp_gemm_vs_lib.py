"""The three item-layer products of the wide Mult-VAE: csrc/gemm.hip vs the vendor library behind torch.mm (rocBLAS /
hipBLASLt, fp32) on the same operands — a calibration of the hand-written kernel, nothing of it is linked into the product."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd._lib import call
from neurec_amd.engine import _ptr, _stream

B, h, I = 512, 600, 40981
r = lambda *s: torch.randn(*s, device="cuda")
ws = torch.empty(64 * B * h * 4, dtype=torch.uint8, device="cuda")
torch.backends.cuda.matmul.allow_tf32 = False


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


def mine(A, lda, Bm, ldb, M, N, K, C, ldc, sp=1):
    call("nrhip_gemm_kmajor", _ptr(A), lda, _ptr(Bm), ldb, M, N, K, _ptr(C), ldc, 0, None, -1, sp, _ptr(ws),
         ws.numel() if sp > 1 else 0, _stream())


fl = 2.0 * B * h * I
g, W, D = r(B, h), r(h, I), r(B, I)
gT, DT, WT = g.t().contiguous(), D.t().contiguous(), W.t().contiguous()
S, dW, dg = r(B, I), r(h, I), r(B, h)
rows = []
t = timed(lambda: mine(gT, B, W, I, B, I, h, S, I)); rows.append(("logits = g W        ", "gemm.hip (A = g^T given)", t))
t = timed(lambda: torch.mm(g, W, out=S)); rows.append(("logits = g W        ", "torch.mm", t))
t = timed(lambda: mine(g, h, D, I, h, I, B, dW, I)); rows.append(("dW = g^T D          ", "gemm.hip", t))
t = timed(lambda: torch.mm(gT, D, out=dW)); rows.append(("dW = g^T D          ", "torch.mm (g^T given)", t))
t = timed(lambda: torch.mm(g.t(), D, out=dW)); rows.append(("dW = g^T D          ", "torch.mm (transposed view)", t))
t = timed(lambda: mine(DT, B, WT, h, B, h, I, dg, h, 32)); rows.append(("dg = D W^T          ", "gemm.hip (D^T, W^T given; split 32)", t))
t = timed(lambda: torch.mm(D, WT, out=dg)); rows.append(("dg = D W^T          ", "torch.mm (W^T given)", t))
t = timed(lambda: torch.mm(D, W.t(), out=dg)); rows.append(("dg = D W^T          ", "torch.mm (transposed view)", t))
for name, who, t in rows:
    print("%s %-40s %7.1f us  %6.1f TFLOP/s" % (name, who, t, fl / t / 1e6))
