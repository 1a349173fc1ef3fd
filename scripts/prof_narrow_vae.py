"""Mult-VAE at the shipped p_dim [16, 32], gowalla shape: a few steps for rocprofv3 --kernel-trace --stats."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import MultiVAEEngine
from neurec_amd.util.tool import get_initializer

train, _ = synth.interactions("gowalla", seed=2018)
U, I = train.shape
wi = get_initializer("xavier_normal", 0.01, seed=2017)
bi = get_initializer("tnormal", 0.01, seed=2018)
z, h, B = 16, 32, 512
params = {"Wq0": wi([I, h]), "bq0": bi([h]), "Wq1": wi([h, 2 * z]), "bq1": bi([2 * z]), "Wp0": wi([z, h]),
          "bp0": bi([h]), "Wp1t": np.ascontiguousarray(wi([h, I]).T), "bp1": bi([I])}
vae = MultiVAEEngine(E.DeviceCSR.from_scipy(train), I, params, 0.001, 0.0, "tanh", B)
perm = torch.from_numpy(np.random.RandomState(0).permutation(U).astype(np.int32)).cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for k in range(n):
    vae.step(perm[k * B:(k + 1) * B].contiguous(), 0.2, 0.8)
torch.cuda.synchronize()
print("loss", vae.loss())
