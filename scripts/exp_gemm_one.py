"""one shape of scripts/exp_gemm.py, a few launches (for counter passes): python scripts/exp_gemm_one.py logits|dW|dg"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd._lib import call
from neurec_amd.engine import _ptr, _stream
B, h, I = 512, 600, 40981
r = lambda *s: torch.randn(*s, device="cuda")
ws = torch.empty(64 * B * h * 4, dtype=torch.uint8, device="cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "logits"
if which == "logits":
    A, lda, Bm, ldb, M, N, K, sp = r(h, B), B, r(h, I), I, B, I, h, 1
elif which == "dW":
    A, lda, Bm, ldb, M, N, K, sp = r(B, h), h, r(B, I), I, h, I, B, 1
else:
    A, lda, Bm, ldb, M, N, K, sp = r(I, B), B, r(I, h), h, B, h, I, 32
Cm = r(M, N)
for _ in range(5):
    call("nrhip_gemm_kmajor", _ptr(A), lda, _ptr(Bm), ldb, M, N, K, _ptr(Cm), N, 0, None, -1, sp, _ptr(ws),
         ws.numel() if sp > 1 else 0, _stream())
torch.cuda.synchronize()
