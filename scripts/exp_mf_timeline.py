"""Per-workgroup phase timeline of mf_fused_step_kernel (needs a library built with -DNR_MF_TIMELINE:
scripts/exp_mf_timeline.sh).  Wave 0 of each workgroup stamps wall_clock64 (100 MHz)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth
from neurec_amd._lib import lib
from neurec_amd.trainer import BprEpochSampler, MFEngine

tr, _ = synth.interactions("gowalla", seed=2018)
U, I = tr.shape
trc = E.DeviceCSR.from_scipy(tr)
rs = np.random.RandomState(2017)
P0, Q0 = (rs.randn(U, 64) * 0.01).astype(np.float32), (rs.randn(I, 64) * 0.01).astype(np.float32)
sampler = BprEpochSampler(trc, I, batch_size=512, seed=2018, plan_users=U)
mu, mp, mn, plans = sampler.epoch_stream()
eng = MFEngine(P0, Q0, 0.001, 0.0, 512, lazy_period=16)
losses = torch.zeros(400, 2, device="cuda")
eng.run_batches(mu[:512 * 300], mp[:512 * 300], mn[:512 * 300], 512, losses, plans[:3 * 512 * 300])
lib.nrhip_mf_timeline.argtypes = [C.c_void_p]
occ = 96                                            # 1536 occurrences / 16
for k in range(300, 304):                           # single steps WITH a next plan: every workgroup of the launch stamps
    sl = slice(512 * k, 512 * (k + 1))
    eng.step(mu[sl], mp[sl], mn[sl], losses[0], plan=plans[3 * 512 * k:3 * 512 * (k + 1)],
             next_plan=plans[3 * 512 * (k + 1):3 * 512 * (k + 2)])
    out = (C.c_ulonglong * (1024 * 8))()
    assert lib.nrhip_mf_timeline(out) == 0
    t = np.frombuffer(out, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
    n_blocks = 96 + (1536 + (U + I + 15) // 16 + 15) // 16
    t0 = t[:n_blocks, 0].min()
    us = lambda x: (x - t0) * 0.01
    o, m = t[:occ], t[occ:n_blocks]
    print("occurrence wgs : start %.2f..%.2f | keys in +%.2f | rows+gradient +%.2f | run sums+Adam+store +%.2f | loss tail +%.2f (max %.2f) | end %.2f..%.2f us"
          % (us(o[:, 0]).min(), us(o[:, 0]).max(), (o[:, 1] - o[:, 0]).mean() * .01, (o[:, 2] - o[:, 1]).mean() * .01,
             (o[:, 3] - o[:, 2]).mean() * .01, (o[:, 4] - o[:, 3]).mean() * .01, (o[:, 4] - o[:, 3]).max() * .01,
             us(o[:, 4]).min(), us(o[:, 4]).max()))
    st_all = us(t[:n_blocks, 0])
    print("  start by blockIdx %% 8 (XCD): %s" % " ".join("%.2f" % st_all[x::8].mean() for x in range(8)))
    print("  start of blocks 0..15: %s" % " ".join("%.2f" % v for v in st_all[:16]))
    print("  end   by blockIdx %% 8 (XCD): %s" % " ".join("%.2f" % us(np.maximum(t[:n_blocks, 3], t[:n_blocks, 4]))[x::8].max() for x in range(8)))
    print("maintenance wgs: start %.2f..%.2f | wave 0 end %.2f..%.2f us (mean %.2f)"
          % (us(m[:, 0]).min(), us(m[:, 0]).max(), us(m[:, 3]).min(), us(m[:, 3]).max(), us(m[:, 3]).mean()))
