"""BPR-MF step time (gowalla shape, B=512, d=64) under the lazy-Adam replay bound `period`."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import BprEpochSampler, MFEngine
tr, _ = synth.interactions("gowalla", seed=2018)
U, I = tr.shape
trc = E.DeviceCSR.from_scipy(tr)
smp = BprEpochSampler(trc, I, batch_size=512, seed=2018, plan_users=U)
batches = [b for b in smp.batches() if b[0].numel() == 512][:600]
loss = torch.zeros(2, device="cuda")
rs = np.random.RandomState(2017)
P0, Q0 = (rs.randn(U, 64) * 0.01).astype(np.float32), (rs.randn(I, 64) * 0.01).astype(np.float32)
for period in (0, 4, 8, 16, 32, 64):
    mf = MFEngine(P0, Q0, 0.001, 0.0, 512, lazy=period > 0, lazy_period=max(period, 1))
    for b in batches[:300]:
        mf.step(b[0], b[1], b[2], loss, plan=b.plan, next_plan=b.next_plan)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in batches[300:]:
        mf.step(b[0], b[1], b[2], loss, plan=b.plan, next_plan=b.next_plan)
    torch.cuda.synchronize()
    print("period %4d (0 = sweep): %.2f us/step" % (period, (time.perf_counter() - t0) / 300 * 1e6))
