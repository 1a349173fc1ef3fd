"""Wide NGCF (64 / [64, 64, 64]) at the gowalla shape: a few steps, for rocprofv3 --kernel-trace --stats."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth
from neurec_amd.graph import ngcf_adjacency, transpose_csr
from neurec_amd.ngcf_wide import NGCFWideEngine
from neurec_amd.util.tool import get_initializer

train, _ = synth.interactions("gowalla", seed=2018)
U, I = train.shape
A = ngcf_adjacency(train, "norm")
w = get_initializer("xavier_normal", 0.01, seed=2018)
e = get_initializer("xavier_normal", 0.01, seed=2017)
table = np.concatenate([e([U, 64]), e([I, 64])])
weights = [(w([64, 64]), w([1, 64]), w([64, 64]), w([1, 64])) for _ in range(3)]
B = 512
eng = NGCFWideEngine(A, transpose_csr(A), U, I, table, weights, 0.001, 0.0, 0.1, B)
rng = np.random.RandomState(0)
coo = train.tocoo()
loss = torch.zeros(2, device="cuda")
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    pick = rng.randint(0, coo.nnz, B)
    u = torch.from_numpy(coo.row[pick].astype(np.int32)).cuda()
    p = torch.from_numpy(coo.col[pick].astype(np.int32)).cuda()
    n = torch.from_numpy(rng.randint(0, I, B).astype(np.int32)).cuda()
    eng.step(u, p, n, loss)
torch.cuda.synchronize()
print("loss", loss.cpu().numpy())
