"""GPU: BPR-MF step time (gowalla shape, B = 512, d = 64): all-rows sweep / two-launch lazy / one-launch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import BprEpochSampler, MFEngine

tr, _ = synth.interactions("gowalla", seed=2018)
U, I = tr.shape
trc = E.DeviceCSR.from_scipy(tr)
rs = np.random.RandomState(2017)
P0, Q0 = (rs.randn(U, 64) * 0.01).astype(np.float32), (rs.randn(I, 64) * 0.01).astype(np.float32)
sampler = BprEpochSampler(trc, I, batch_size=512, seed=2018, plan_users=U)
batches = [b for b in sampler.batches() if b[0].numel() == 512][:600]
loss = torch.zeros(2, device="cuda")
out = {}
for name, kw in (("sweep", dict(lazy=False)), ("two-launch lazy", dict(lazy=True, fused=False)),
                 ("one-launch", dict(lazy=True, fused=True))):
    for period in ((16,) if name == "sweep" else (4, 8, 16, 32)):
        eng = MFEngine(P0, Q0, 0.001, 0.0, 512, lazy_period=period, **kw)
        for b in batches[:100]:
            eng.step(b[0], b[1], b[2], loss, plan=b.plan, next_plan=b.next_plan)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in batches[100:]:
            eng.step(b[0], b[1], b[2], loss, plan=b.plan, next_plan=b.next_plan)
        host = (time.perf_counter() - t0) / 500 * 1e6          # enqueue rate of the host loop
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 500 * 1e6
        out[(name, period)] = eng.P.cpu().numpy()
        print("%-16s period %2d : %.2f us/step (host enqueue %.2f us/step)" % (name, period, us, host), flush=True)
# the same with the batch loop in C (MFEngine.run_batches): not bound by the Python enqueue rate
sampler2 = BprEpochSampler(trc, I, batch_size=512, seed=2018, plan_users=U)
mu, mp, mn, mplans = sampler2.epoch_stream()
losses = torch.zeros(600, 2, device="cuda")
for name, kw in (("sweep", dict(lazy=False)), ("two-launch lazy", dict(lazy=True, fused=False)),
                 ("one-launch", dict(lazy=True, fused=True))):
    for period in ((16,) if name == "sweep" else (8, 16)):
        eng = MFEngine(P0, Q0, 0.001, 0.0, 512, lazy_period=period, **kw)
        eng.run_batches(mu[:51200], mp[:51200], mn[:51200], 512, losses, mplans[:3 * 51200])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.run_batches(mu[51200:307200], mp[51200:307200], mn[51200:307200], 512, losses, mplans[3 * 51200:3 * 307200])
        host = (time.perf_counter() - t0) / 500 * 1e6
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 500 * 1e6
        out[(name + " / native loop", period)] = eng.P.cpu().numpy()
        print("%-16s period %2d, batch loop in C : %.2f us/step (host enqueue %.2f us/step)" % (name, period, us, host), flush=True)
ref = out[("sweep", 16)]
for k, v in out.items():
    print(k, "bit-identical to the sweep:", bool(np.array_equal(v, ref)))
