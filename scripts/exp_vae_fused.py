"""Fused Mult-VAE decoder at B = 512, h = 32 and I = 8192·k items (k item tiles per workgroup on 256 CUs):
fixed cost and per-tile cost of the two passes (run under rocprofv3 --kernel-trace --stats per k).
    python scripts/exp_vae_fused.py k [batch]"""
import os
import sys
import numpy as np
import scipy.sparse as sp
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E

k = float(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
I, h = int(8192 * k), 32
rng = np.random.RandomState(0)
R = sp.random(B, I, 27.0 / I, random_state=1, format="csr", dtype=np.float32)
R.data[:] = 1.0
R.sort_indices()
csr = E.DeviceCSR.from_scipy(R)
rows = torch.arange(B, dtype=torch.int32, device="cuda")
G1 = torch.from_numpy((rng.randn(B, h) * 0.7).astype(np.float32)).cuda()
W = torch.from_numpy((rng.randn(I, h) * 0.5).astype(np.float32)).cuda()
b = torch.from_numpy((rng.randn(I) * 0.3).astype(np.float32)).cuda()
z = lambda *s: torch.zeros(*s, dtype=torch.float32, device="cuda")
out = (z(B), z(I, h), z(I), z(B, h))
ws = E.vae_fused_workspace(B, I, "cuda")
for _ in range(20):
    E.vae_decoder_fused(I, b, csr, rows, G1, W, *out, ws)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    E.vae_decoder_fused(I, b, csr, rows, G1, W, *out, ws)
e.record()
torch.cuda.synchronize()
print("I = %d (%.2f tiles per workgroup), B = %d: decoder %.1f us" % (I, I / 32 / 256, B, a.elapsed_time(e) * 20))
