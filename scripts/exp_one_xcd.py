"""a persistent BPR-MF epoch kernel confined to ONE XCD (<= 32 workgroups, barrier inside one L2).
Two measurements decide it before any such kernel is written:
  (a) the per-step barrier among G = 8 / 16 / 32 workgroups of one XCD (one counter, release add + acquire spin);
  (b) the step's WORK on one XCD: the shipped one-launch step (nrhip_mf_steps) on a CU-masked stream that only
      reaches one XCD's 32 CUs — no barrier, no launch saved yet; the persistent form cannot be faster than this.
Config 2 (BASELINE configs[1]): gowalla shape, d = 64, B = 512."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib  # noqa: E402

lib = _explib.load()
p, i32 = C.c_void_p, C.c_int
lib.nrhip_exp_xcd_barrier.argtypes = [p, i32, i32, i32, p, p]
lib.nrhip_exp_xcc_histogram.argtypes = [i32, p, p]
lib.nrhip_exp_cumask_stream_create.argtypes = [p, i32, C.POINTER(p)]
lib.nrhip_exp_cumask_stream_destroy.argtypes = [p]
cur = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

cnt = torch.zeros(16, dtype=torch.int32, device="cuda")
out = torch.zeros(4096, device="cuda")


def barrier_us(group, iters):
    def run(n):
        assert lib.nrhip_exp_xcd_barrier(cnt.data_ptr(), 0, group, n, out.data_ptr(), cur()) == 0
    run(10)
    torch.cuda.synchronize()
    ts = []
    for n in (1, iters + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(n); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    void = int(cnt[2].item())
    return (ts[1] - ts[0]) / iters, ts[0], void


print("(a) barrier among G workgroups of ONE XCD, us per barrier (2,000 barriers; 'void' = a workgroup never arrived)")
for g in (8, 16, 32):
    us, base, void = barrier_us(g, 2000)
    print("    G = %2d   %.2f us per barrier   (one-barrier kernel %.1f us)%s" % (g, us, base, "   VOID" if void else ""))

# ---- (b) the shipped step on one XCD's CUs
hist = torch.zeros(8, dtype=torch.int32, device="cuda")


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << (b - 32 * w) for b in bits if 32 * w <= b < 32 * (w + 1)) for w in range(8)])
    h = p(0)
    assert lib.nrhip_exp_cumask_stream_create(words, 8, C.byref(h)) == 0
    return h


def xcds_of(handle):
    ext = torch.cuda.ExternalStream(handle.value)
    with torch.cuda.stream(ext):
        assert lib.nrhip_exp_xcc_histogram(4096, hist.data_ptr(), cur()) == 0
    ext.synchronize()
    return hist.cpu().numpy()


cands = {"bits k with k % 8 == 0": [k for k in range(256) if k % 8 == 0], "bits 0..31": list(range(32))}
chosen = None
for name, bits in cands.items():
    h = masked_stream(bits)
    hh = xcds_of(h)
    print("CU mask %-24s -> blocks per XCD %s" % (name, hh.tolist()))
    if (hh > 0).sum() == 1 and chosen is None:
        chosen = (name, h)
    else:
        lib.nrhip_exp_cumask_stream_destroy(h)
print("(b) BPR-MF step (config 2: gowalla shape, d = 64, B = 512), one launch per step, batch loop in C:")
from neurec_amd import engine as E, synth  # noqa: E402
from neurec_amd.trainer import BprEpochSampler, MFEngine  # noqa: E402

train, _ = synth.interactions("gowalla", seed=2018)
U, I = train.shape
trc = E.DeviceCSR.from_scipy(train)
rs = np.random.RandomState(2017)


def mf_us(stream_ctx):
    mf = MFEngine((rs.randn(U, 64) * 0.01).astype(np.float32), (rs.randn(I, 64) * 0.01).astype(np.float32), 0.001, 0.0, 512)
    sam = BprEpochSampler(trc, I, neg_num=1, batch_size=512, shuffle=True, seed=2018, plan_users=U)
    mu, mp_, mn, plans = sam.epoch_stream()
    loss = torch.zeros(400, 2, device="cuda")
    torch.cuda.synchronize()
    with stream_ctx:
        mf.run_batches(mu[:50 * 512], mp_[:50 * 512], mn[:50 * 512], 512, loss, plans[:3 * 50 * 512])
        torch.cuda.current_stream().synchronize()
        t0 = time.perf_counter()
        mf.run_batches(mu[50 * 512:450 * 512], mp_[50 * 512:450 * 512], mn[50 * 512:450 * 512], 512, loss,
                       plans[3 * 50 * 512:3 * 450 * 512])
        torch.cuda.current_stream().synchronize()
    return (time.perf_counter() - t0) / 400 * 1e6


import contextlib  # noqa: E402
print("    whole chip (256 CUs)                      %.2f us per step" % mf_us(contextlib.nullcontext()))
if chosen is not None:
    ext = torch.cuda.ExternalStream(chosen[1].value)
    print("    one XCD (32 CUs; mask: %s)   %.2f us per step" % (chosen[0], mf_us(torch.cuda.stream(ext))))
else:
    print("    no CU mask selected exactly one XCD: not measured")
