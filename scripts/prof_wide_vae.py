"""Wide Mult-VAE (p_dim [200, 600]) at the gowalla shape: a few steps, for rocprofv3 --kernel-trace --stats.
usage (through gpurun): bash scripts/prof_wide_vae.sh"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth
from neurec_amd.util.tool import get_initializer
from neurec_amd.vae_wide import MultiVAEWideEngine

B, zw, hw = 512, 200, 600
train, _ = synth.interactions("gowalla", seed=2018)
U, I = train.shape
wi = get_initializer("xavier_normal", 0.01, seed=2017)
bi = get_initializer("tnormal", 0.01, seed=2018)
dev = E.require_gpu()
eng = MultiVAEWideEngine(E.DeviceCSR.from_scipy(train), I, [wi([I, hw]), wi([hw, 2 * zw])], [bi([hw]), bi([2 * zw])],
                         [wi([zw, hw]), wi([hw, I])], [bi([hw]), bi([I])], 0.001, 0.0, "tanh", B)
perm = torch.from_numpy(np.random.RandomState(0).permutation(U).astype(np.int32)).to(dev)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    eng.step(perm[k * B:(k + 1) * B].contiguous(), 0.2, 0.8)
torch.cuda.synchronize()
print("loss", eng.loss())
