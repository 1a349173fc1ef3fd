"""The legs of bench.py — one function per BASELINE configuration next to the headline (configs[2], timed in
bench.main itself): leg_mf (configs[0] ml-100k and configs[1] gowalla BPR-MF), leg_ngcf / leg_multivae (configs[4]),
leg_config4 + _config4_eval + leg_config4_partitions (configs[3]: training, evaluation, one rank's share of eight),
cpu_baseline (SURVEY 8d's three CPU legs — the only place, with leg_mf's `with_cpu` branch, that touches `oracle/`),
and compact_line (the ONE line the driver parses).  Split out of bench.py in round 6; nothing
here runs inside the headline's timed region."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth of the 8 XCDs (same guide, section "L2 (per XCD)")
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 matrix peak (same guide); v_mfma_f32_32x32x2_f32
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak (same guide; AMD's 2:1-sparse headline figure is not used)
MFMA_I8_PEAK_TOPS = 5000.0      # dense int8 matrix rate: the guide gives no spec row, "~2x the bf16 rate" (ubench >= 3,944)


def _hip_timed(fn, n, warm=3):
    """average milliseconds of fn() over n calls, HIP events on torch's current stream (the launch stream)"""
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def _cpu_timed(step, budget=6.0, max_steps=8):
    """(seconds per step, steps timed): one warm call, then up to max_steps inside the budget"""
    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    if warm > budget:
        return warm, 1
    t0, n = time.perf_counter(), 0
    while n < max_steps and time.perf_counter() - t0 < budget:
        step()
        n += 1
    return (time.perf_counter() - t0) / n, n


def _median_wall(fn, runs=5):
    """(median seconds, last result) of whole host-side calls bracketed by device synchronisation"""
    import torch
    ts, out = [], None
    for _ in range(runs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], out


def leg_mf(train, test, trc, tec, dev, eval_batch, eval_mode, with_eval, with_cpu, label="gowalla"):
    """BPR-MF (MF.py:45-113 with conf/MF.properties: d = 64, B = 512, lr 0.001, reg 0, N(0, 0.01) tables) on the given
    interactions: BASELINE configs[1] at the gowalla shape, configs[0] at the ml-100k shape (the reference's own
    CPU-runnable case).  A step = ONE launch: gather, BPR forward/backward, the batch's duplicate-row sums, TF-1.12 sparse
    Adam by exact lazy replay (bit-identical to the all-rows sweep).  One whole epoch is timed (sampler and
    batch plans inside, the short last batch too) next to a window of full steps.  -> (info, engine)"""
    import torch
    from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, MFEngine
    U, I = train.shape
    B, d = 512, 64
    rs = np.random.RandomState(2017)
    P0, Q0 = (rs.randn(U, d) * 0.01).astype(np.float32), (rs.randn(I, d) * 0.01).astype(np.float32)
    mf = MFEngine(P0, Q0, 0.001, 0.0, B)                                      # conf/MF.properties
    mf_sampler = BprEpochSampler(trc, I, neg_num=1, batch_size=B, shuffle=True, seed=2018, plan_users=U)
    # the batch loop of MF.train_model runs natively (MFEngine.run_batches -> nrhip_mf_steps): a
    # Python loop enqueues ~12 us per step, about what the one-launch step takes on the GPU
    mu, mp, mn, mplans = mf_sampler.epoch_stream()
    avail = mu.numel() // B
    w_steps = min(50, avail // 4)
    t_steps = max(min(400, avail - w_steps), 1)
    n_warm, n_timed = w_steps * B, t_steps * B
    mf_loss = torch.zeros(max(avail + 1, 1), 2, device=dev)
    cut = lambda lo, hi: (mu[lo:hi], mp[lo:hi], mn[lo:hi])
    mf.run_batches(*cut(0, n_warm), B, mf_loss, mplans[:3 * n_warm])
    # the window in three parts, the median part reported (it is ~1-5 ms of wall clock: one stall would be the whole
    # figure); the same steps in the same order as one window, so the tables the evaluation below sees are the same
    win, part = [], max(t_steps // 3, 1)
    for lo_s in range(0, t_steps, part):
        hi_s = t_steps if lo_s + 2 * part > t_steps else lo_s + part
        lo, hi = n_warm + lo_s * B, n_warm + hi_s * B
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mf.run_batches(*cut(lo, hi), B, mf_loss, mplans[3 * lo:3 * hi])
        torch.cuda.synchronize()
        win.append((time.perf_counter() - t0) / (hi_s - lo_s))
        if hi_s == t_steps:
            break
    mf_dt = sorted(win)[len(win) // 2]
    # ONE WHOLE EPOCH as MF.train_model runs it (MF.py:95-103): the sampler's launch, the batch plans, every batch of
    # the permuted stream including the short last one — E / epoch wall time, SURVEY 8d's metric
    ep = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eu_, ep_, en_, epl = mf_sampler.epoch_stream()
        mf.run_batches(eu_, ep_, en_, B, mf_loss, epl)
        torch.cuda.synchronize()
        ep.append(time.perf_counter() - t0)
    epoch_s = sorted(ep)[1]
    n_epoch = int(mu.numel())
    # the same steps with TF's literal all-rows sweep (the checker) for the record
    mf_sweep = MFEngine(P0, Q0, 0.001, 0.0, B, lazy=False)
    mf_sweep.run_batches(*cut(0, n_warm), B, mf_loss, mplans[:3 * n_warm])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mf_sweep.run_batches(*cut(n_warm, n_warm + n_timed), B, mf_loss, mplans[3 * n_warm:3 * (n_warm + n_timed)])
    torch.cuda.synchronize()
    sweep_dt = (time.perf_counter() - t0) / t_steps
    del mf_sweep
    touched = B * (72 * d + 12)              # 3 rows x (read + write of p, m, v) + ids, per triplet
    info = {"shape": label, "users": U, "items": I, "interactions": int(train.nnz),
            "triplets_per_sec": B / mf_dt, "ms_per_step": mf_dt * 1e3, "batch": B, "dim": d, "steps_timed": t_steps,
            "epoch": {"triplets_per_sec": n_epoch / epoch_s, "ms": epoch_s * 1e3, "triplets": n_epoch,
                      "steps": (n_epoch + B - 1) // B, "runs_ms": [x * 1e3 for x in ep],
                      "note": "sampler launch + batch plans + every batch (short last one included), wall clock between "
                              "device synchronisations, median of 3"},
            "optimizer": "TF-1.12 sparse Adam by exact lazy replay (bit-identical to the all-rows sweep), "
                         "gradient + optimiser in one launch on double-buffered tables",
            "roofline": {"bound": "hbm", "bytes_per_step": touched, "unit": "GB/s",
                         "achieved": touched / mf_dt / 1e9, "peak": HBM_PEAK_GBS,
                         "frac": touched / mf_dt / 1e9 / HBM_PEAK_GBS,
                         "epoch_frac": n_epoch * (72 * d + 12) / epoch_s / 1e9 / HBM_PEAK_GBS,
                         "note": "SURVEY 8d bound (72 d + 12) B per triplet; one launch whose critical path is a "
                                 "chain of ~5 dependent memory round trips (plan key -> ids -> stamps + rows -> "
                                 "ordered row sums -> Adam -> store -> loss reduction): latency-bound, not "
                                 "bandwidth-bound"},
            "sweep_ms_per_step": sweep_dt * 1e3,
            "sweep_GBps": 2 * 4 * (U + I) * d * 4 / sweep_dt / 1e9}
    tu = None
    if with_eval:
        # configs[0] / [1] name the evaluator too: the BPR-MF tables through the same full-rank path
        tu = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).to(dev)
        mf_ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=eval_batch, pruned=eval_mode == "pruned")
        mf_ev.evaluate_factors(mf.P, mf.Q, tu)
        mf_edt, mm = _median_wall(lambda: mf_ev.evaluate_factors(mf.P, mf.Q, tu))
        info["eval"] = {"users_per_sec": tu.numel() / mf_edt, "ms": mf_edt * 1e3, "n_users": int(tu.numel()),
                        "ndcg@10": float(mm[2 * 20 + 9]), "search": getattr(mf_ev, "search_used", None),
                        "rows_redone": int(getattr(mf_ev, "n_flagged", 0)),
                        "mfma_fp32_roof_ratio": 2.0 * I * d * tu.numel() / mf_edt / 1e12 / MFMA_F32_PEAK_TFLOPS}
    if with_cpu:
        # the CPU side of configs[0]: the restated MF step (oracle.train.mf_step: numpy gathers + TF's all-rows sparse
        # Adam, 1 thread — TF-CPU itself is not installable here) and the reference's own C++ evaluator fed by
        # np.matmul as MF.py:120-122 does, on the GPU run's tables and users
        from oracle import native, ref, train as otrain
        Pc, Qc = P0.copy(), Q0.copy()
        mP, vP, mQ, vQ = (np.zeros_like(x) for x in (Pc, Pc, Qc, Qc))
        adam = otrain.Adam(0.001)
        hu, hp, hn = (x.cpu().numpy() for x in (mu, mp, mn))
        t0, n = time.perf_counter(), 0
        while n < avail and time.perf_counter() - t0 < 6.0:
            sl = slice(n * B, (n + 1) * B)
            otrain.mf_step(Pc, Qc, mP, vP, mQ, vQ, hu[sl], hp[sl], hn[sl], 0.0, adam)
            n += 1
        cpu_step = (time.perf_counter() - t0) / max(n, 1)
        cb = {"value": B / cpu_step, "unit": "triplets/s", "cores": 1, "kind": "port",
              "sample": "%d BPR-MF steps (B = %d, d = %d) of the same stream: oracle.train.mf_step, numpy fp32, the "
                        "all-rows sparse Adam of TF 1.12" % (n, B, d)}
        if tu is not None:
            users = tu.cpu().numpy()[:1024]
            Ph, Qh = mf.P.cpu().numpy(), mf.Q.cpu().numpy()
            truth = [test.indices[test.indptr[u]:test.indptr[u + 1]].tolist() for u in users]
            fn = ref.eval_matrix if ref.available() else native.eval_matrix
            res = []
            t0 = time.perf_counter()
            for b in range(0, len(users), 128):               # test_batch_size=128, NeuRec.properties:40
                ub = users[b:b + 128]
                S = np.ascontiguousarray(np.matmul(Ph[ub], Qh.T), dtype=np.float32)
                native.mask_train(S, ub, train.indptr.astype(np.int64), train.indices)
                res.append(fn(S, truth[b:b + 128], [1, 2, 4, 3, 5], 20, threads=8))
            dte = time.perf_counter() - t0
            want = float(np.mean(np.concatenate(res), axis=0)[2 * 20 + 9])
            got = mf_ev.evaluate_factors(mf.P, mf.Q, torch.from_numpy(users).to(dev), exact_mean=True)
            cb["eval"] = {"value": len(users) / dte, "unit": "users/s", "cores": 8,
                          "kind": "reference" if ref.available() else "port", "ndcg@10": want,
                          "sample": "%d users, np.matmul + C++ evaluator, num_thread=8, batch 128" % len(users)}
            info["ndcg10_oracle_absdiff"] = abs(float(got[2 * 20 + 9]) - want)
        info["cpu_baseline"] = cb
    return info, mf


def leg_ngcf(train, test, trc, tec, dev, with_cpu):
    """BASELINE configs[4], NGCF half (conf/NGCF.properties: embedding 16, layers [16, 16], B = 512, `norm`
    adjacency, message dropout 0.1) on the run's interactions: step time, the roofline of its dominant
    kernel (the 16-wide SpMM: four passes per step) and the oracle.train port timed on the host."""
    import torch
    from neurec_amd.graph import ngcf_adjacency, transpose_csr
    from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, NGCFEngine
    from neurec_amd.util.tool import get_initializer
    U, I = train.shape
    A = ngcf_adjacency(train, "norm")
    At = transpose_csr(A)
    w = get_initializer("xavier_normal", 0.01, seed=2018)
    e = get_initializer("xavier_normal", 0.01, seed=2017)
    table = np.concatenate([e([U, 16]), e([I, 16])])
    weights = [(w([16, 16]), w([1, 16]), w([16, 16]), w([1, 16])) for _ in range(2)]
    B, lr, reg, drop = 512, 0.001, 0.0, 0.1
    ng = NGCFEngine(A, At, U, I, table, weights, lr, reg, drop, B)
    sampler = BprEpochSampler(trc, I, neg_num=1, batch_size=B, shuffle=True, seed=2018, plan_users=U)
    batches = [b for b in sampler.batches() if b[0].numel() == B][:200]
    loss = torch.zeros(2, device=dev)
    it = iter(batches * 10)

    def step():
        b = next(it)
        ng.step(b[0], b[1], b[2], loss, plan=b.plan)
    ms = _hip_timed(step, 150, 20)
    # dominant kernel: S = A·E at d = 16 (lane-group kernel, 4 lanes per row), 2 forward + 2 backward per step
    x, y = ng.ego[0], ng.S[0]
    spmm_ms = _hip_timed(lambda: ng.A.matmul(x, out=y), 40, 5)
    spmm_bytes = ng.A.algorithmic_bytes(16)
    users = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).to(dev)
    ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=32768)   # (as the headline leg: --eval-batch default)

    def evaluate():
        eu, ei = ng.final_embeddings()
        return ev.evaluate_factors(eu.contiguous(), ei.contiguous(), users)
    evaluate()
    edt, m = _median_wall(evaluate)
    out = {"ms_per_step": ms, "triplets_per_sec": B / ms * 1e3, "batch": B, "dim": 16, "layers": [16, 16],
           "adjacency_nnz": int(A.nnz), "launches": "one native call per step (nrhip_ngcf_step: ~27 launches)",
           "roofline": {"bound": "hbm", "kernel": ng.A.full_pass_kernel(16), "bytes_per_launch": spmm_bytes,
                        "us_per_launch": spmm_ms * 1e3, "launches_per_step": 4,
                        "achieved": spmm_bytes / spmm_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": spmm_bytes / spmm_ms / 1e6 / HBM_PEAK_GBS, "traffic": None,
                        "note": "nnz*8 + (N+1)*4 + 2*N*16*4 B per pass (SURVEY 8d's SpMM formula at d = 16); the "
                                "4.5 MB operand table is L2-sized, the pass is bound by the CSR stream and latency"},
           "eval": {"users_per_sec": users.numel() / edt, "ms": edt * 1e3, "ndcg@10": float(m[2 * 20 + 9]),
                    "rows_redone": int(getattr(ev, "n_flagged", 0)), "search": getattr(ev, "search_used", None)}}
    # the NGCF paper's widths (embedding 64, layers [64, 64, 64]) on the width-generic engine
    from neurec_amd.ngcf_wide import NGCFWideEngine
    table64 = np.concatenate([e([U, 64]), e([I, 64])])
    weights64 = [(w([64, 64]), w([1, 64]), w([64, 64]), w([1, 64])) for _ in range(3)]
    wide = NGCFWideEngine(A, At, U, I, table64, weights64, lr, reg, drop, B)
    it2 = iter(batches * 10)

    def wide_step():
        b = next(it2)
        wide.step(b[0], b[1], b[2], loss, plan=b.plan)
    wide_ms = _hip_timed(wide_step, 40, 8)
    x64, y64 = wide.ego[0], wide.S[0]
    spmm64_ms = _hip_timed(lambda: wide.A.matmul(x64, out=y64), 40, 5)
    spmm64_bytes = wide.A.algorithmic_bytes(64)
    out["wide"] = {"dim": 64, "layers": [64, 64, 64], "batch": B, "ms_per_step": wide_ms,
                   "triplets_per_sec": B / wide_ms * 1e3,
                   "roofline": {"bound": "hbm", "kernel": wide.A.full_pass_kernel(64), "bytes_per_launch": spmm64_bytes,
                                "us_per_launch": spmm64_ms * 1e3, "launches_per_step": 6,
                                "achieved": spmm64_bytes / spmm64_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": spmm64_bytes / spmm64_ms / 1e6 / HBM_PEAK_GBS, "traffic": None,
                                "note": "3 forward + 3 backward SpMM passes at d = 64 per step; the dense layer products "
                                        "run on the fp32 matrix cores (csrc/gemm.hip); one native call per step "
                                        "(nrhip_ngcf_wide_step: ~70 launches; r03 issued them from Python: 1.02 ms)"}}
    del wide
    if with_cpu:
        from oracle import train as O
        rng = np.random.RandomState(3)
        coo = train.tocoo()
        params = [table.copy()] + [x_.copy() for ws in weights for x_ in ws]
        ms_, vs_ = [np.zeros_like(p) for p in params], [np.zeros_like(p) for p in params]
        adam = O.Adam(lr)

        def cpu_step():
            pick = rng.randint(0, coo.nnz, B)
            masks = [(rng.rand(U + I, 16) < 1 - drop).astype(np.float32) for _ in weights]
            Wl = [tuple(params[1 + 4 * k:5 + 4 * k]) for k in range(2)]
            _, dE, wg = O.ngcf_loss_and_grads(A, At, params[0], Wl, masks, 1 - drop, U, coo.row[pick],
                                              coo.col[pick], rng.randint(0, I, B), reg)
            for p_, m_, v_, g_ in zip(params, ms_, vs_, [dE] + [x_ for gs in wg for x_ in gs]):
                adam.dense(p_, m_, v_, g_.reshape(p_.shape))
            adam.advance()
        sec, n = _cpu_timed(cpu_step)
        out["cpu_baseline"] = {"value": B / sec, "unit": "triplets/s", "cores": 1, "kind": "port",
                               "sample": "%d NGCF steps (B=%d) of oracle.train (scipy CSR SpMM + numpy fp32, 1 thread; "
                                         "pinned to the reference's NGCF class by tests/test_tfgraph_golden.py)" % (n, B)}
    return out


def leg_multivae(train, test, trc, tec, dev, with_cpu):
    """BASELINE configs[4], Mult-VAE half (conf/MultiVAE.properties: p_dim [16, 32], B = 512, tanh, keep 0.8)."""
    import torch
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator, MultiVAEEngine
    from neurec_amd.util.tool import get_initializer
    U, I = train.shape
    wi = get_initializer("xavier_normal", 0.01, seed=2017)
    bi = get_initializer("tnormal", 0.01, seed=2018)
    z, h, B = 16, 32, 512
    params = {"Wq0": wi([I, h]), "bq0": bi([h]), "Wq1": wi([h, 2 * z]), "bq1": bi([2 * z]), "Wp0": wi([z, h]),
              "bp0": bi([h]), "Wp1t": np.ascontiguousarray(wi([h, I]).T), "bp1": bi([I])}
    vae = MultiVAEEngine(trc, I, params, 0.001, 0.0, "tanh", B)
    perm = torch.from_numpy(np.random.RandomState(0).permutation(U).astype(np.int32)).to(dev)
    rows_list = [perm[k * B:(k + 1) * B].contiguous() for k in range(U // B)]
    it = iter(rows_list * 20)
    ms = _hip_timed(lambda: vae.step(next(it), 0.2, 0.8, want_loss=True), 150, 20)
    # dominant kernels: the decoder's loss + gradients with NO [B][I] buffer (csrc/vae_fused.hip): logits tiles are
    # recomputed on the fp32 matrix cores in two passes (statistics; gradients)
    rows = rows_list[0]
    vae.step(rows, 0.2, 0.8)

    def decoder():
        E.vae_decoder_fused(I, vae.P["bp1"], vae.csr, rows, vae.G1[:B], vae.P["Wp1t"], vae.nll[:B],
                            vae.G["Wp1t"], vae.G["bp1"], vae.dG1[:B], vae.ws)
    dec_ms = _hip_timed(decoder, 40, 5)
    for k in vae.G:
        vae.G[k].zero_()
    nnz_b = int((vae.csr.h_indptr[rows.cpu().numpy().astype(np.int64) + 1]
                 - vae.csr.h_indptr[rows.cpu().numpy().astype(np.int64)]).sum())
    # algorithmic bytes: W_p1 + b_p1 read, dW_p1 + db_p1 written, g1 read, dg1 + nll written, the batch's CSR rows
    dec_bytes = 2 * (I * h + I) * 4 + (2 * B * h + B) * 4 + nnz_b * 4 + 2 * B * 8
    # flops issued: the logits twice (K = h + the bias step, 17 MFMA steps of 2) and the two gradients once each
    dec_flops = 2.0 * B * I * (2 * (h + 2) + 2 * h)
    dec_flops_min = 3 * 2.0 * B * I * h                    # logits, dW_p1 = G^T g1, dg1 = G W_p1 once each
    users = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).to(dev)
    ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=32768)   # (as the headline leg: --eval-batch default)

    def evaluate():
        pf, qf = vae.eval_factors()
        return ev.evaluate_factors(pf, qf, users)
    evaluate()
    edt, m = _median_wall(evaluate)
    out = {"ms_per_step": ms, "users_per_sec_train": B / ms * 1e3, "batch": B, "p_dim": [z, h],
           "roofline": {"bound": "mfma", "kernel": "vae_dec_stats_kernel + vae_dec_grad_kernel<2> (+ rows / stat / dg1-reduce: "
                                                   "nrhip_vae_decoder_fused)",
                        "flops_per_launch": dec_flops, "us_per_launch": dec_ms * 1e3, "launches_per_step": 1,
                        "achieved": dec_flops / dec_ms / 1e9, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": dec_flops / dec_ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                        "frac_of_minimal_flops": dec_flops_min / dec_ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                        "algorithmic_bytes": dec_bytes, "hbm_frac_on_algorithmic_bytes": dec_bytes / dec_ms / 1e6 / HBM_PEAK_GBS,
                        "traffic": None,
                        "note": "no [B][I] logits slab any more: the 32-deep logits product is recomputed tile by tile in both "
                                "passes (4 GEMM-equivalents of 2·B·I·h issued where 3 are the minimum), so the bound is the fp32 "
                                "matrix pipe, not HBM: algorithmic bytes are W_p1 + dW_p1 + the batch rows (10.6 MB, ~1.3 us at "
                                "8 TB/s).  frac = issued flops / time / dense fp32 MFMA peak; frac_of_minimal_flops counts the "
                                "recomputation as waste.  The slab form it replaces moved 336 MB (profiles/r04_narrow_vae_*)"},
           "eval": {"users_per_sec": users.numel() / edt, "ms": edt * 1e3, "ndcg@10": float(m[2 * 20 + 9]),
                    "design": "factor path: logits = [g1(u) | 1]·[W_p1 | b_p1] through the pruned evaluator"}}
    # conf/MultiVAE.properties:3's alternative shape p_dim = [200, 600] on the width-generic engine
    from neurec_amd.vae_wide import MultiVAEWideEngine
    zw, hw = 200, 600
    wide = MultiVAEWideEngine(trc, I, [wi([I, hw]), wi([hw, 2 * zw])], [bi([hw]), bi([2 * zw])],
                              [wi([zw, hw]), wi([hw, I])], [bi([hw]), bi([I])], 0.001, 0.0, "tanh", B)
    it2 = iter(rows_list * 20)
    wide_ms = _hip_timed(lambda: wide.step(next(it2), 0.2, 0.8, want_loss=True), 60, 10)
    wide.step(rows, 0.2, 0.8)

    def item_layer():                                      # logits, dW = g^T D, dg = D W^T — every operand as it lies
        wide._gemm(wide.Gp[-1], hw, 1, wide.Wp[-1], I, 0, B, I, hw, wide.S, wide.ld, bias=wide.bp[-1])
        wide._gemm(wide.Gp[-1], hw, 0, wide.S, wide.ld, 0, hw, I, B, wide.G[2 * 2 + 1], I)
        wide._gemm(wide.S, wide.ld, 1, wide.Wp[-1], I, 1, B, hw, I, wide.dGp[-1], hw, splits=wide.splits)
    item_ms = _hip_timed(item_layer, 30, 5)
    item_flops = 3 * 2.0 * B * I * hw
    out["wide"] = {"p_dim": [zw, hw], "batch": B, "ms_per_step": wide_ms, "users_per_sec_train": B / wide_ms * 1e3,
                   "roofline": {"bound": "mfma", "kernel": "gemm_lds_kernel<128> x3 (+ 1 split reduce): the item layer's logits, dW "
                                                           "and dg, every operand read in the layout it is stored in",
                                "flops_per_step": item_flops, "us_per_step": item_ms * 1e3,
                                "achieved": item_flops / item_ms / 1e9, "peak": MFMA_F32_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": item_flops / item_ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                                "traffic": None,
                                "note": "fp32 (the reference's dtype) on v_mfma_f32_32x32x2_f32: 128 x 128 block tiles, "
                                        "buffer-loaded operand tiles (k-major or k-minor) double-buffered through LDS, blocks dealt "
                                        "to the XCDs by n-tile; the vendor library on the same products: 75-108 TFLOP/s "
                                        "(profiles/r03_exp_gemm_fp32_mfma.txt)"}}
    del wide
    if with_cpu:
        from oracle import train as O
        rng = np.random.RandomState(4)
        p = {k: v.copy() for k, v in params.items()}
        p["Wp1"] = np.ascontiguousarray(p.pop("Wp1t").T)
        names = ("Wq0", "bq0", "Wq1", "bq1", "Wp0", "bp0", "Wp1", "bp1")
        plist = [p[k] for k in names]
        ms_, vs_ = [np.zeros_like(x_) for x_ in plist], [np.zeros_like(x_) for x_ in plist]
        adam = O.Adam(0.001)

        def cpu_step():
            rws = rng.choice(U, B, replace=False)
            X = np.asarray(train[rws].todense(), dtype=np.float32)            # MultiVAE.py:152-165
            mask = (rng.rand(B, I) < 0.8).astype(np.float32)
            eps = (rng.randn(B, z) * 0.01).astype(np.float32)
            _, (gWq, gbq, gWp, gbp), _ = O.multivae_loss_and_grads(
                X, [p["Wq0"], p["Wq1"]], [p["bq0"], p["bq1"]], [p["Wp0"], p["Wp1"]], [p["bp0"], p["bp1"]],
                mask, np.float32(0.8), eps, 0.2, 0.0, "tanh")
            for p_, m_, v_, g_ in zip(plist, ms_, vs_, [gWq[0], gbq[0], gWq[1], gbq[1], gWp[0], gbp[0], gWp[1], gbp[1]]):
                adam.dense(p_, m_, v_, g_.reshape(p_.shape))
            adam.advance()
        sec, n = _cpu_timed(cpu_step)
        try:
            nthreads = len(os.sched_getaffinity(0))
        except AttributeError:
            nthreads = os.cpu_count() or 1
        out["cpu_baseline"] = {"value": B / sec, "unit": "users/s", "cores": nthreads, "kind": "port",
                               "sample": "%d Mult-VAE steps (B=%d) of oracle.train (numpy fp32; its [512 x 40981] "
                                         "matmuls run on the BLAS threads of the box; pinned to the reference's "
                                         "MultiVAE class by tests/test_tfgraph_golden.py)" % (n, B)}
    return out


def leg_config4(comm, dev, scale, batch=8192, dim=128, layers=3, steps=3, hop=None, eval_users=0, eval_batch=65536):
    """BASELINE configs[3] (LightGCN, U = 10^7, I = 10^6, E = 2*10^8, d = 128) at `scale` on this GPU through
    the row-sharded engine: graph generated on the device, adjacency block built on the device.  hop: the form of the
    per-hop exchange (sharded.ShardedLightGCN: sliced / allgather / chunked / reduce; None = the engine's default)."""
    import torch
    from neurec_amd import engine as E, parallel as par, synth
    from neurec_amd.sharded import ShardedLightGCN
    from neurec_amd.trainer import BprEpochSampler
    t_setup = time.perf_counter()
    U, I, n_edges = (max(int(x * scale), 64) for x in synth.CONFIG4)
    tr_ptr, tr_idx = synth.device_interactions(U, I, n_edges, seed=2018, device=dev)
    n_train = int(tr_ptr[-1])
    part = par.BipartitePartition(U, I, comm.world)
    ur, ir = part.users_of(comm.rank), part.items_of(comm.rank)
    rows = synth.device_lightgcn_rank_rows(tr_ptr, tr_idx, U, I, ur, ir)
    lim = float(np.sqrt(6.0 / (U + I + dim)))
    g = torch.Generator(device=dev)
    g.manual_seed(2017 + comm.rank)
    E0 = (torch.rand((ur[1] - ur[0]) + (ir[1] - ir[0]), dim, generator=g, device=dev) * 2 - 1) * lim
    lg = ShardedLightGCN(comm, None, U, I, E0, layers, 0.01, 1e-3, batch, local_rows=rows, hop=hop)
    del rows, E0
    trc = E.DeviceCSR(tr_ptr, tr_idx, I)
    sampler = BprEpochSampler(trc, I, neg_num=1, batch_size=batch, shuffle=True, seed=2018, rank=comm.rank,
                              world=comm.world, plan_users=None)
    it = sampler.batches()
    first = next(it)
    lg.plan_epoch(sampler._users[:sampler.n_local], sampler._pos[:sampler.n_local], sampler._neg[:sampler.n_local],
                  batch)
    bs = [first] + [next(it) for _ in range(steps)]
    lg.step(bs[0][0], bs[0][1], bs[0][2], None, batch_index=0)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup
    comm.barrier()
    t0 = time.perf_counter()
    for k in range(1, steps + 1):
        lg.step(bs[k][0], bs[k][1], bs[k][2], None, batch_index=k)
    torch.cuda.synchronize()
    comm.barrier()
    dt = comm.max_float(time.perf_counter() - t0) / steps

    def hops():
        for k in range(2 * layers):
            lg.local_pass(k)
    spmm_ms = _hip_timed(hops, 2, 1) / (2 * layers)          # per HOP: every launch of the form, no collective
    spmm_bytes = lg.A.algorithmic_bytes(dim)
    gathered = int(lg.A.nnz) * dim * 4
    exchange = None
    if comm.live:
        # one whole hop with the run's own collectives (the form's exchange overlapped the way the step overlaps it)
        comm.barrier()
        hop_ms = _hip_timed(lambda: lg._hops((lg.A, lg.R), [dict(src=lg.E0, out=lg.Ya, addend=lg.H),
                                                             dict(src=lg.Ya, out=lg.Yb, addend=lg.H)]), 3, 1) / 2
        recv = (comm.world - 1) * 4 * dim * (2 * part.bi if lg.hop == "reduce" else part.b)
        exchange = {"hop_form": lg.hop, "column_slabs": lg.S, "hop_ms": comm.max_float(hop_ms),
                    "hop_compute_only_ms": spmm_ms, "hops_per_step": 2 * layers,
                    "received_per_rank_bytes_per_hop": int(recv),
                    "lookup_all_to_all_bytes_per_step": int(3 * batch * dim * 4 * 3), "backend": comm.backend,
                    "note": {"sliced": "the table lives as column slabs; slab s+1 is all-gathered (every link busy) under "
                                       "the one-launch SpMM of slab s, across hop boundaries too; exact",
                             "allgather": "one all-gather of the [b][d] blocks, then the one-launch SpMM; exact",
                             "chunked": "r04: W rank-ordered broadcasts under W carry launches; exact",
                             "reduce": "user rows: all-gather of the item blocks only; item rows: per-rank partials over "
                                       "the rank's own user rows, equal-split all-to-all, rank-ordered sums at the "
                                       "owners; within fp32 rounding of the exact forms"}[lg.hop] +
                            "; ids -> rows -> gradient rows by three all-to-alls"}
    out = {"scale": scale, "users": U, "items": I, "interactions": n_train, "dim": dim, "batch": batch,
           "layers": layers, "steps": steps, "ms_per_step": dt * 1e3, "triplets_per_sec": comm.world * batch / dt,
           "setup_seconds": setup_s, "ranks": comm.world, "hop": lg.hop, "exchange": exchange,
           "roofline": {"bound": "hbm", "kernel": lg.A.full_pass_kernel(lg.w), "bytes_per_launch": spmm_bytes,
                        "us_per_launch": spmm_ms * 1e3, "launches_per_step": 2 * layers,
                        "achieved": spmm_bytes / spmm_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": spmm_bytes / spmm_ms / 1e6 / HBM_PEAK_GBS, "traffic": None,
                        "row_gather_bytes_per_launch": gathered, "row_gather_GBps": gathered / spmm_ms / 1e6,
                        "note": "per hop (all %d slab launches); algorithmic bytes read every operand row once; a CSR "
                                "pass gathers one %d-B row per non-zero from a table no cache holds — random row "
                                "gathers run at 7.3-7.4 TB/s on this part whether the table sits in the Infinity "
                                "Cache or in HBM (6.3 at 5 GB; profiles/r03_exp_gather_vs_table_size.txt), which is "
                                "the rate the pass sustains" % (lg.S, lg.w * 4)}}
    if eval_users:
        try:
            out["eval"] = _config4_eval(lg, comm, trc, I, dim, dev, eval_users, batch_rows=eval_batch)
        except Exception as e:                                # a secondary leg must not take the headline down
            out["eval"] = {"error": "%s: %s" % (type(e).__name__, e)}
    del lg, sampler, trc, tr_ptr, tr_idx
    torch.cuda.empty_cache()
    return out


def _config4_eval(lg, comm, trc, I, dim, dev, n_eval, batch_rows=8192, top_k=20):
    """The evaluation half of BASELINE's metric at the config-4 shape (uni_evaluator.py:101-157 on LightGCN.py:183-192):
    every rank ranks ITS users against the whole item table — ShardedLightGCN.eval_factors all-gathers the item blocks
    only — and the metric sums are added over the ranks (sharded.ShardedEvaluator).  Timed on a BOUNDED block of each
    rank's users (the first n_eval of them: ranking is per user, the rate does not depend on which), the propagation
    that precedes an evaluation timed on its own (it is per evaluation, not per user)."""
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.sharded import ShardedEvaluator
    mids = [1, 2, 4, 3, 5]
    n = min(int(n_eval), lg.nu)
    train_rows = trc.rows(lg.ulo, lg.ulo + n)
    test_rows = synth.device_test_rows(train_rows, I, per_user=2, seed=2019 + comm.rank)
    ev = ShardedEvaluator(comm, train_rows, test_rows, mids, top_k, batch_rows=batch_rows)
    torch.cuda.synchronize(); comm.barrier()
    t0 = time.perf_counter()
    eu, items = lg.eval_factors()
    torch.cuda.synchronize(); comm.barrier()
    factors_s = comm.max_float(time.perf_counter() - t0)
    eu = eu[:n]
    t0 = time.perf_counter()
    ev.evaluate_factors(eu, items)                            # first call: strike plan, filter operands, buffers
    torch.cuda.synchronize()
    first_s = time.perf_counter() - t0
    runs = []
    for _ in range(3):
        torch.cuda.synchronize(); comm.barrier()
        t0 = time.perf_counter()
        means = ev.evaluate_factors(eu, items)
        torch.cuda.synchronize(); comm.barrier()
        runs.append(comm.max_float(time.perf_counter() - t0))
    dt = sorted(runs)[1]
    fe = ev.ev
    out = {"users_per_sec": ev.n_total / dt, "ms": dt * 1e3, "ms_runs": [r * 1e3 for r in runs], "n_users": ev.n_total,
           "users_per_rank": ev.n_local, "items": I, "dim": dim, "batch_rows": batch_rows,
           "ndcg@10": float(means[2 * top_k + 9]), "search": getattr(fe, "search_used", None), "rows_redone": ev.rows_redone,
           "first_call_ms": first_s * 1e3, "factors_ms": factors_s * 1e3,
           "whole_population_seconds": factors_s + lg.n_users / (ev.n_total / dt),
           "sample": "the first %d users of every rank with a synthetic test split (2 uniform draws per user outside the "
                     "train row); factors_ms = the L hops + the all-gather of the item blocks that precede an "
                     "evaluation (per evaluation, not per user); whole_population_seconds = factors + U / users_per_sec; "
                     "the model is %d steps from Xavier noise: NDCG@10 is at chance, the leg measures the rate" % (n, 4)}
    # roofline of the search (the dominant kernel) and of the ranking phase, first batch, HIP events on the launch stream
    ub = ev.users[:batch_rows]
    nb = ub.numel()
    search = getattr(fe, "search_used", None)                  # None: too few tiles for the pruned path (tiny tables)
    filt = fe._filter if search in ("bf16", "int8") else None
    if nb and search is not None and fe._plan is not None:
        n_keep = min(top_k + 1 + fe.extra_tiles + (fe.int8_extra_tiles if search == "int8" else 0), 63) \
            if filt is not None else top_k + 1
        level1 = lambda: fe._gemm.tile_maxima(eu, ub, train_rows, plan=fe._plan, row_of=fe._row_of, filt=filt)
        r1 = level1()
        M, eps = r1 if filt is not None else (r1, None)
        per = torch.empty((nb, len(mids) * top_k), dtype=torch.float32, device=dev)
        flg = torch.zeros(nb, dtype=torch.int32, device=dev)
        level2 = lambda: E.eval_tiles(M, eu, fe._gemm, ub, train_rows, test_rows, mids, top_k, per, flg, eps=eps,
                                      n_keep=n_keep if eps is not None else None)
        t1 = _hip_timed(level1, 3, 1) * 1e-3
        t2 = _hip_timed(level2, 3, 1) * 1e-3
        flops = 2.0 * I * dim * nb
        if filt is not None:
            i8 = filt.arith == "int8"
            peak, sustained = (MFMA_I8_PEAK_TOPS, 3500.0) if i8 else (MFMA_BF16_PEAK_TFLOPS, 1300.0)
            out["roofline"] = {"bound": "mfma", "arith": filt.arith, "users": nb, "ms": t1 * 1e3,
                               "kernel": ("tilemax_i8" if i8 else "tilemax_bf16") + ("_wide_kernel" if dim > 64 else "_kernel"),
                               "achieved": 3.0 * flops / t1 / 1e12, "peak": peak, "unit": "TOP/s" if i8 else "TFLOP/s",
                               "frac": 3.0 * flops / t1 / 1e12 / peak, "frac_of_sustained": 3.0 * flops / t1 / 1e12 / sustained,
                               "note": "operations ISSUED (3 products of 2·I·d per user) over split + filter + planned fix-up"}
        else:
            out["roofline"] = {"bound": "mfma", "arith": "fp32", "users": nb, "ms": t1 * 1e3, "kernel": "score_tilemax_kernel",
                               "achieved": flops / t1 / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": flops / t1 / 1e12 / MFMA_F32_PEAK_TFLOPS}
        tiles = 2 * ((I + 63) // 64)
        # level 2, algorithmic HBM bytes: per user its tile maxima read once, its factor row, M·K metrics; the rescored
        # item tiles read once per 32 (user, tile) pairs of a bucket chunk
        rank_bytes = nb * (tiles * 4 + dim * 4 + len(mids) * top_k * 4) + nb * n_keep * 32 * dim * 4 // 32
        out["roofline_topk"] = {"bound": "hbm", "ms": t2 * 1e3, "bytes": rank_bytes, "achieved": rank_bytes / t2 / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rank_bytes / t2 / 1e9 / HBM_PEAK_GBS,
                                "tile_maxima_per_user": tiles, "tiles_rescored_per_user": n_keep,
                                "kernel": "select_rows_kernel (streaming ring over %d maxima) + tile_count / tile_fill "
                                          "(packed buckets) + rescore_pairs_kernel + rank_compact_kernel (ranking, certificate, metrics)" % tiles}
    return out


def leg_config4_partitions(dev, scale, which="both", W=8, dim=128, L=3, B=8192):
    """BASELINE configs[3] on ONE GPU: what ONE rank of a W-rank job computes per step under the two partitions, at the
    config-4 law —
      rowshard_rank0_of_W  1/W of the users and of the items, all `dim` columns; per hop the other blocks arrive in
                           rank-ordered chunks under the launches (sharded.ChunkedHop) — here the chunk buffers are
                           resident and nothing is exchanged: the compute side of the hop, chunked and one-launch;
      colshard_rank0_of_W  all nodes, dim/W columns, the global batch of W·B (colshard.py): no per-hop exchange."""
    import torch
    from neurec_amd import engine as E, parallel as par, synth
    U, I, n_edges = (max(int(x * scale), 64) for x in synth.CONFIG4)
    tr_ptr, tr_idx = synth.device_interactions(U, I, n_edges, seed=2018, device=dev)
    out = {"scale": scale, "users": U, "items": I, "interactions": int(tr_ptr[-1]), "dim": dim, "ranks_modelled": W}

    class Share(par.Comm):
        """rank 0 of W with every exchange a no-op: the compute share of one rank on one GPU"""
        def __init__(self):
            super().__init__(0, W, 0, "none")

        def barrier(self):
            pass

        def bcast_rows_start(self, buf, src):
            return None

        def all_gather_rows(self, local, out_):
            return out_

        def all_gather_rows_start(self, local, out_):
            return None

        def all_to_all_equal_start(self, send, recv):
            return None

        def _done(self, work=None):
            pass

    if which in ("rows", "both"):
        from neurec_amd.sharded import ShardedLightGCN
        part = par.BipartitePartition(U, I, W)
        ur, ir = part.users_of(0), part.items_of(0)
        rows = synth.device_lightgcn_rank_rows(tr_ptr, tr_idx, U, I, ur, ir)
        lim = float(np.sqrt(6.0 / (U + I + dim)))
        E0 = (torch.rand((ur[1] - ur[0]) + (ir[1] - ir[0]), dim, device=dev) * 2 - 1) * lim
        b, forms = part.b, {}
        nnz = None
        for name, kw in (("allgather", dict(hop="allgather")), ("sliced2", dict(hop="sliced", col_slices=2)),
                         ("sliced4", dict(hop="sliced", col_slices=4)), ("reduce", dict(hop="reduce")),
                         ("chunked", dict(hop="chunked"))):
            t0 = time.perf_counter()
            lg = ShardedLightGCN(Share(), None, U, I, E0, L, 0.01, 1e-3, B, local_rows=rows, **kw)
            torch.cuda.synchronize()
            build_s = time.perf_counter() - t0
            nnz = int(lg.A.nnz)
            for k in range(min(lg.S, 2)):
                lg._xbuf(k).uniform_(-lim, lim)             # what the (skipped) all-gather would have delivered
            hop = [dict(src=lg.E0, out=lg.Ya, addend=lg.H)]
            f = {"build_seconds": build_s, "column_slabs": lg.S, "slab_bytes_per_row": lg.w * 4,
                 "hop_compute_ms": _hip_timed(lambda: lg._hops((lg.A, lg.R), hop), 3, 1),
                 "kernel": lg.A.full_pass_kernel(lg.w),
                 "received_bytes_per_hop": (W - 1) * 4 * dim * (2 * part.bi if name == "reduce" else b)}
            if name == "reduce":
                Mu, Mp = lg.R
                lg._Z.uniform_(-lim, lim)
                bu = part.bu
                f["partial_product_ms"] = _hip_timed(lambda: Mp.matmul(lg.E0[0], out=lg._P), 3, 1)
                f["user_rows_ms"] = _hip_timed(lambda: Mu.matmul(lg._Z, out=lg.Ya[0][:bu], addend=lg.H[0][:bu]), 3, 1)
                f["ordered_sum_of_%d_partials_ms" % W] = _hip_timed(
                    lambda: E.partials_sum_rows(lg._Rv.view(W, b - bu, dim), W, out=lg.Ya[0][bu:], addend=lg.H[0][bu:]), 3, 1)
                f["nnz_user_rows"], f["nnz_partial"] = int(Mu.nnz), int(Mp.nnz)
            if name == "chunked":
                ch = lg.A.chunked
                Yv, slots = ch.buffers(dim, dev)
                per = []
                for r in range(W):                          # the chunk launches one by one
                    c = ch.chunks[r]
                    per.append(_hip_timed(lambda: E.call("nrhip_spmm_csr_carry", c.plan, E._ptr(c.indptr), E._ptr(c.indices),
                                                         E._ptr(c.vals), E._ptr(lg.E0[0]), E._ptr(slots[r & 1]), b, dim,
                                                         E._ptr(Yv), 1 if r else 0, E._ptr(None, allow_none=True),
                                                         E._stream()), 3, 1))
                f["chunk_launch_ms"], f["virtual_rows"] = per, ch.n_virtual
                del ch, Yv, slots
            f["row_gather_GBps"] = nnz * dim * 4 / f["hop_compute_ms"] / 1e6
            forms[name] = f
            del lg
            torch.cuda.empty_cache()
        out["rowshard_rank0_of_%d" % W] = {"rows_per_rank": b, "nnz_per_rank": nnz, "row_gather_bytes_per_hop": nnz * dim * 4,
                                           "hops_per_step": 2 * L, "forms": forms}
        del rows, E0
        torch.cuda.empty_cache()
    if which in ("cols", "both"):
        from neurec_amd.trainer import LightGCNEngine
        N = U + I
        t0 = time.perf_counter()
        ip, idx, val = synth.device_lightgcn_rank_rows(tr_ptr, tr_idx, U, I, (0, U), (0, I))   # every row, natural ids
        A = E.SpmmCSR(ip, idx, val, n_cols=N, split_row=U)
        del ip, idx, val
        gB, dl = W * B, dim // W
        lim = float(np.sqrt(6.0 / (N + dim)))
        emb = (torch.rand(N, dl, device=dev) * 2 - 1) * lim           # rank 0's columns
        lgc = LightGCNEngine(A, U, I, emb, L, 0.01, 1e-3, gB)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        trc = E.DeviceCSR(tr_ptr, tr_idx, I)
        n_b = 3
        users, pos, neg = E.sample_bpr_epoch(trc, trc.row_of(), I, 1, 2018, 0, True, begin=0, count=n_b * gB)
        parts = torch.zeros(3 * gB, device=dev)
        ctx, k = lgc._ctx, [0]

        def step():
            j = k[0] % n_b
            k[0] += 1
            bu, bp, bn = (t[j * gB:(j + 1) * gB] for t in (users, pos, neg))
            ctx.lightgcn_step_colshard_fwd(bu, bp, bn, parts)
            ctx.lightgcn_step_colshard_bwd(bu, bp, bn, lgc.adam, None, None, parts)
            lgc.adam.advance()
        ms = _hip_timed(step, 3, 1)
        ms_hop = _hip_timed(lambda: lgc.A.matmul(lgc.E0, out=lgc.Ea, addend=lgc.H), 3, 1)
        out["colshard_rank0_of_%d" % W] = {
            "columns_per_rank": dl, "kernel_width": lgc.d, "global_batch": gB, "build_seconds": build_s, "ms_per_step": ms,
            "hop_ms": ms_hop, "kernel": lgc.A.full_pass_kernel(lgc.d), "nnz": int(A.nnz),
            "row_gather_bytes_per_hop": int(A.nnz) * dl * 4, "row_gather_GBps": int(A.nnz) * dl * 4 / ms_hop / 1e6,
            "exchange_bytes_per_rank_per_step": 12 * gB, "triplets_per_sec_if_exchange_were_free": gB / ms * 1e3}
        del lgc, A, trc
        torch.cuda.empty_cache()
    # what the measured compute sides mean at W ranks over xGMI (7 links x 76.8 GB/s per direction into a GPU) — ONE
    # number per form, under ONE stated assumption; the chunked form's second number is there because its
    # collective is W one-source broadcasts and nothing measured here says they use more than the direct link
    link_in = 7 * 76.8e9
    rs, cs = out.get("rowshard_rank0_of_%d" % W), out.get("colshard_rank0_of_%d" % W)
    model = {"link_in_bytes_per_s": link_in,
             "assumed": "all_gather_into_tensor / all_to_all_single keep all 7 incoming links busy; steady state of a "
                        "chain of hops (the first slab of a step's first hop is exposed once)"}
    if rs:
        fm = rs["forms"]
        ms = lambda f: f["received_bytes_per_hop"] / link_in * 1e3
        model["hop_ms"] = {
            "allgather": ms(fm["allgather"]) + fm["allgather"]["hop_compute_ms"],
            "sliced2": max(ms(fm["sliced2"]), fm["sliced2"]["hop_compute_ms"]),
            "sliced4": max(ms(fm["sliced4"]), fm["sliced4"]["hop_compute_ms"]),
            "reduce": max(ms(fm["reduce"]) / 2, fm["reduce"]["partial_product_ms"]) +
            max(ms(fm["reduce"]) / 2, fm["reduce"]["user_rows_ms"]) + fm["reduce"]["ordered_sum_of_%d_partials_ms" % W],
            "chunked": max(ms(fm["chunked"]), sum(fm["chunked"]["chunk_launch_ms"])) + fm["chunked"]["chunk_launch_ms"][-1] +
            (fm["chunked"]["hop_compute_ms"] - sum(fm["chunked"]["chunk_launch_ms"])),
            "chunked_if_broadcasts_use_one_link": fm["chunked"]["received_bytes_per_hop"] / 76.8e9 * 1e3 +
            fm["chunked"]["chunk_launch_ms"][-1]}
        model["hop_bound_by"] = {k: ("links" if ms(fm[k]) > fm[k]["hop_compute_ms"] else "compute")
                                 for k in ("sliced2", "sliced4")}
        model["hop_bound_by"]["reduce"] = "links" if ms(fm["reduce"]) / 2 > min(fm["reduce"]["partial_product_ms"],
                                                                                 fm["reduce"]["user_rows_ms"]) else "compute"
        model["step_ms"] = {k: 2 * L * v for k, v in model["hop_ms"].items()}
    if cs:
        model["colshard_step_ms"] = cs["ms_per_step"]
    out["model_at_%d_ranks" % W] = model
    return out


def cpu_baseline(train, test, E0, args, eval_tables=None, n_eval_users=1024):
    """SURVEY §8d's CPU legs, timed on this box's host cores on bounded samples of the same workload:
      (i)   the reference's own PairwiseSampler epoch (data/sampler.py + util/data_iterator.py +
            util/cython/random_choice.pyx compiled as they are into oracle/_ref; 1 Python thread);
      (ii)  the LightGCN step port (oracle.train: scipy CSR SpMM + numpy, 1 thread) AND its
            torch-CPU twin at torch.set_num_threads(nproc) (oracle.train_torch) — `value` is the
            faster of the two, with the cores it used;
      (iii) the reference's own C++ evaluator (oracle/_ref, num_thread=8, test_batch_size=128) fed by
            np.matmul as MF.py:120-122 does, on `eval_tables` (the GPU run's E* tables, so that its
            NDCG@10 can be compared with the GPU evaluator's on the same users)."""
    from oracle import native, ref, train as otrain
    from oracle.train_torch import TorchLightGCN
    U, I = train.shape
    coo = train.tocoo()
    A = otrain.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
    B = args.batch
    rows = np.repeat(np.arange(U), np.diff(train.indptr))
    rng = np.random.RandomState(1)
    picks = [(rng.randint(0, train.nnz, B), rng.randint(0, I, B)) for _ in range(args.cpu_steps + 1)]
    try:
        nproc = len(os.sched_getaffinity(0))
    except AttributeError:
        nproc = os.cpu_count() or 1

    def time_steps(step, budget=8.0):
        """(triplets/s, steps timed): up to cpu_steps steps, stopped early once `budget` seconds are
        spent (a leg never runs away with the bench's few minutes)."""
        t0 = time.perf_counter()
        step(rows[picks[0][0]], train.indices[picks[0][0]], picks[0][1])        # warm caches
        warm = time.perf_counter() - t0
        if warm > budget:
            return B / warm, 1
        t0, n = time.perf_counter(), 0
        for pick, neg in picks[1:]:
            step(rows[pick], train.indices[pick], neg)
            n += 1
            if time.perf_counter() - t0 > budget:
                break
        return n * B / (time.perf_counter() - t0), n
    E = E0.copy()
    m, v = np.zeros_like(E), np.zeros_like(E)
    adam = otrain.Adam(0.01)
    scipy_1t, n_scipy = time_steps(lambda u, p, n: otrain.lightgcn_step(A, A, E, m, v, U, args.layers, u,
                                                                        p, n, 1e-3, adam))
    import torch
    before = torch.get_num_threads()
    torch_nt, torch_threads, n_torch = 0.0, nproc, 0
    for threads in sorted({nproc, min(nproc, 32), min(nproc, 8)}, reverse=True):
        # torch.set_num_threads(nproc) as SURVEY §8d says; fewer threads are tried too because a
        # sparse-CSR SpMM of this size does not scale to hundreds of threads — the best is reported
        rate, n = time_steps(TorchLightGCN(A, E0, U, args.layers, 0.01, 1e-3, threads).step, budget=5.0)
        if rate > torch_nt:
            torch_nt, torch_threads, n_torch = rate, threads, n
    torch.set_num_threads(before)
    best_torch = torch_nt > scipy_1t
    out = {"value": max(scipy_1t, torch_nt), "unit": "triplets/s",
           "cores": torch_threads if best_torch else 1, "kind": "port",
           "sample": "LightGCN steps (B=%d, L=%d, d=%d) of the same graph; faster of scipy CSR SpMM + "
                     "numpy fp32 on 1 thread (%d steps, %.0f triplets/s) and torch-CPU sparse-CSR at its "
                     "best thread count %d of %d available (%d steps, %.0f triplets/s)"
                     % (B, args.layers, args.dim, n_scipy, scipy_1t, torch_threads, nproc, n_torch, torch_nt),
           "step_scipy_1thread": scipy_1t, "step_torch_best": torch_nt, "torch_threads": torch_threads,
           "host_cores_available": nproc}
    # (i) sampler leg: the reference's own code when oracle/_ref travelled with the snapshot
    mod = ref.sampler_module()
    if mod is not None:
        class _Dataset:                                   # what PairwiseSampler reads (sampler.py:191-192)
            num_items = I

            @staticmethod
            def get_user_train_dict():
                return {u: train.indices[train.indptr[u]:train.indptr[u + 1]].tolist()
                        for u in range(U) if train.indptr[u + 1] > train.indptr[u]}
        np.random.seed(2018)                              # main.py:10
        smp = mod.PairwiseSampler(_Dataset, neg_num=1, batch_size=B, shuffle=True)
        t0 = time.perf_counter()
        n = 0
        for bu, _, _ in smp:
            n += len(bu)
        out["sampler"] = {"value": n / (time.perf_counter() - t0), "unit": "triplets/s", "cores": 1,
                          "kind": "reference",
                          "sample": "one PairwiseSampler epoch (%d triplets, B=%d): the reference's "
                                    "data/sampler.py + util/data_iterator.py + Cython random_choice, "
                                    "compiled unchanged" % (n, B)}
    else:
        out["sampler"] = None                             # oracle/_ref did not travel: not timed
    # (iii) evaluator leg
    users = np.flatnonzero(np.diff(test.indptr) > 0)[:n_eval_users].astype(np.int32)
    P, Q = eval_tables if eval_tables is not None else (E[:U], E[U:])
    truth = [test.indices[test.indptr[u]:test.indptr[u + 1]].tolist() for u in users]
    res = []
    t0 = time.perf_counter()
    for b in range(0, len(users), 128):                  # test_batch_size=128, NeuRec.properties:40
        ub = users[b:b + 128]
        S = np.ascontiguousarray(np.matmul(P[ub], Q.T), dtype=np.float32)
        native.mask_train(S, ub, train.indptr.astype(np.int64), train.indices)
        fn = ref.eval_matrix if ref.available() else native.eval_matrix
        res.append(fn(S, truth[b:b + 128], [1, 2, 4, 3, 5], 20, threads=8))
    dte = time.perf_counter() - t0
    out["eval"] = {"value": len(users) / dte, "unit": "users/s", "cores": 8,
                   "kind": "reference" if ref.available() else "port",
                   "sample": "%d users, np.matmul + C++ evaluator, num_thread=8, batch 128" % len(users),
                   "ndcg@10": float(np.mean(np.concatenate(res), axis=0)[2 * 20 + 9]),
                   "users": users}
    return out


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(line):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline` with the
    secondary legs' headline numbers as FLAT numeric keys (the driver's parser keeps those objects key by key and
    truncates anything long), no prose.  The full objects — every leg with its kernel names, notes and sample
    descriptions — go to bench_full.json (committed per round as profiles/rNN_bench.json)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "rccl_ranks", "dist_backend", "redundant_compute", "final_loss")
    out = {k: line[k] for k in keep if k in line}
    # the metric as SURVEY 8d defines it — one whole timed epoch, sampler and short last batch inside — next to the
    # contract's K-step window
    out["value_epoch"] = _get(line, "epoch_timed", "value")
    out["epoch_ms"] = _get(line, "epoch_timed", "ms")
    out["epoch_steps"] = _get(line, "epoch_timed", "steps")
    out["epoch_vs_steps_window"] = _get(line, "epoch_timed", "vs_steps_window")
    short = lambda d: {k: v for k, v in (d or {}).items()
                       if isinstance(v, (int, float, bool)) or v is None or (isinstance(v, str) and len(v) <= 96)}
    roof = short(line.get("roofline"))
    legs = {
        "epoch_amortised_triplets_per_sec": _get(line, "epoch_amortised", "value"),
        "eval_users_per_sec": _get(line, "eval", "users_per_sec"), "eval_ms": _get(line, "eval", "ms"),
        "eval_ndcg10": _get(line, "eval", "ndcg@10"),
        "eval_ndcg10_oracle_absdiff": _get(line, "eval", "ndcg10_oracle_absdiff"),
        "eval_mfma_tflops": _get(line, "eval", "roofline", "achieved"),
        "eval_mfma_frac": _get(line, "eval", "roofline", "frac"),
        "eval_search": _get(line, "eval", "search"),
        "eval_search_frac_of_sustained": _get(line, "eval", "roofline", "frac_of_sustained"),
        "eval_search_unit": _get(line, "eval", "roofline", "unit"),
        "eval_fp32_loop_ms": _get(line, "eval", "roofline", "fp32_mfma_loop_ms"),
        "eval_fp32_roof_ratio": _get(line, "eval", "fp32_roof_ratio"),
        "eval_search_ms": _get(line, "eval", "roofline", "ms"),
        "eval_rank_ms": _get(line, "eval", "roofline_topk", "ms"),
        "eval_rows_redone": _get(line, "eval", "rows_redone_for_ties"),
        "eval_plan_build_ms": _get(line, "eval", "strike_plan_build_ms"),
        "mf_triplets_per_sec": _get(line, "mf", "triplets_per_sec"),
        "mf_us_per_step": None if _get(line, "mf", "ms_per_step") is None else _get(line, "mf", "ms_per_step") * 1e3,
        "mf_hbm_frac": _get(line, "mf", "roofline", "frac"),
        "mf_eval_users_per_sec": _get(line, "mf", "eval", "users_per_sec"),
        "mf_epoch_triplets_per_sec": _get(line, "mf", "epoch", "triplets_per_sec"),
        "ml100k_mf_triplets_per_sec": _get(line, "ml100k", "epoch", "triplets_per_sec"),
        "ml100k_mf_window_triplets_per_sec": _get(line, "ml100k", "triplets_per_sec"),
        "ml100k_mf_us_per_step": None if _get(line, "ml100k", "ms_per_step") is None else _get(line, "ml100k", "ms_per_step") * 1e3,
        "ml100k_mf_hbm_frac": _get(line, "ml100k", "roofline", "frac"),
        "ml100k_eval_users_per_sec": _get(line, "ml100k", "eval", "users_per_sec"),
        "ml100k_eval_ms": _get(line, "ml100k", "eval", "ms"),
        "ml100k_eval_ndcg10": _get(line, "ml100k", "eval", "ndcg@10"),
        "ml100k_eval_fp32_roof_ratio": _get(line, "ml100k", "eval", "mfma_fp32_roof_ratio"),
        "ml100k_ndcg10_oracle_absdiff": _get(line, "ml100k", "ndcg10_oracle_absdiff"),
        "ngcf_ms_per_step": _get(line, "ngcf", "ms_per_step"),
        "ngcf_triplets_per_sec": _get(line, "ngcf", "triplets_per_sec"),
        "ngcf_spmm_hbm_frac": _get(line, "ngcf", "roofline", "frac"),
        "ngcf_wide_ms_per_step": _get(line, "ngcf", "wide", "ms_per_step"),
        "multivae_ms_per_step": _get(line, "multivae", "ms_per_step"),
        "multivae_users_per_sec": _get(line, "multivae", "users_per_sec_train"),
        "multivae_decoder_us": _get(line, "multivae", "roofline", "us_per_launch"),
        "multivae_decoder_mfma_frac": _get(line, "multivae", "roofline", "frac"),
        "multivae_wide_ms_per_step": _get(line, "multivae", "wide", "ms_per_step"),
        "multivae_wide_mfma_frac": _get(line, "multivae", "wide", "roofline", "frac"),
        "config4_ms_per_step": _get(line, "config4", "ms_per_step"),
        "config4_triplets_per_sec": _get(line, "config4", "triplets_per_sec"),
        "config4_spmm_hbm_frac": _get(line, "config4", "roofline", "frac"),
        "config4_row_gather_GBps": _get(line, "config4", "roofline", "row_gather_GBps"),
        "config4_eval_users_per_sec": _get(line, "config4", "eval", "users_per_sec"),
        "config4_eval_n_users": _get(line, "config4", "eval", "n_users"),
        "config4_eval_search": _get(line, "config4", "eval", "search"),
        "config4_eval_rows_redone": _get(line, "config4", "eval", "rows_redone"),
        "config4_eval_factors_ms": _get(line, "config4", "eval", "factors_ms"),
        "config4_eval_whole_population_s": _get(line, "config4", "eval", "whole_population_seconds"),
        "config4_eval_mfma_tflops": _get(line, "config4", "eval", "roofline", "achieved"),
        "config4_eval_mfma_frac": _get(line, "config4", "eval", "roofline", "frac"),
        "config4_eval_mfma_frac_of_sustained": _get(line, "config4", "eval", "roofline", "frac_of_sustained"),
        "config4_eval_search_ms": _get(line, "config4", "eval", "roofline", "ms"),
        "config4_eval_rank_ms": _get(line, "config4", "eval", "roofline_topk", "ms"),
        "config4_eval_rank_hbm_frac": _get(line, "config4", "eval", "roofline_topk", "frac"),
        "config4_rank0of8_hop_ms_allgather": _get(line, "config4", "partitions", "model_at_8_ranks", "hop_ms", "allgather"),
        "config4_rank0of8_hop_ms_sliced2": _get(line, "config4", "partitions", "model_at_8_ranks", "hop_ms", "sliced2"),
        "config4_rank0of8_hop_ms_sliced4": _get(line, "config4", "partitions", "model_at_8_ranks", "hop_ms", "sliced4"),
        "config4_rank0of8_hop_ms_reduce": _get(line, "config4", "partitions", "model_at_8_ranks", "hop_ms", "reduce"),
        "config4_rank0of8_hop_compute_ms_sliced2": _get(line, "config4", "partitions", "rowshard_rank0_of_8", "forms",
                                                        "sliced2", "hop_compute_ms"),
        "config4_rank0of8_hop_compute_ms_reduce": _get(line, "config4", "partitions", "rowshard_rank0_of_8", "forms",
                                                       "reduce", "hop_compute_ms"),
        "config4_colshard_rank0of8_step_ms": _get(line, "config4", "partitions", "colshard_rank0_of_8", "ms_per_step"),
    }
    for w in ("2", "4", "8"):
        legs["colshard_share_ms_%s" % w] = _get(line, "colshard_one_rank_share", w, "ms_per_step")
    legs["same_global_batch_on_1gpu_triplets_per_sec"] = _get(line, "same_global_batch_on_1gpu", "value")
    legs["exchange_ms_per_step"] = _get(line, "exchange_measured", "ms_per_step")
    legs["strong_scaling_triplets_per_sec"] = _get(line, "strong_scaling", "value")
    legs["strong_scaling_ms_per_step"] = _get(line, "strong_scaling", "ms_per_step")
    legs["rowshard_config4_law_ms_per_step"] = _get(line, "rowshard_config4_law", "ms_per_step")
    legs["rowshard_config4_law_hop"] = _get(line, "rowshard_config4_law", "hop")
    legs["rowshard_config4_law_hop_ms"] = _get(line, "rowshard_config4_law", "exchange", "hop_ms")
    legs["rowshard_config4_law_reduce_ms_per_step"] = _get(line, "rowshard_config4_law_reduce", "ms_per_step")
    legs["rowshard_config4_law_reduce_hop_ms"] = _get(line, "rowshard_config4_law_reduce", "exchange", "hop_ms")
    legs["rowshard_config4_law_eval_users_per_sec"] = _get(line, "rowshard_config4_law", "eval", "users_per_sec")
    legs["rowshard_config4_law_eval_factors_ms"] = _get(line, "rowshard_config4_law", "eval", "factors_ms")
    roof.update({k: v for k, v in legs.items() if v is not None})
    out["roofline"] = roof
    cb = line.get("cpu_baseline")
    if cb is not None:
        c = short(cb)
        for k, path in (("sampler_triplets_per_sec", ("sampler", "value")), ("eval_users_per_sec", ("eval", "value")),
                        ("eval_ndcg10", ("eval", "ndcg@10"))):
            v = _get(cb, *path)
            if v is not None:
                c[k] = v
        for k, path in (("ml100k_mf_triplets_per_sec", ("ml100k", "cpu_baseline", "value")),
                        ("ml100k_eval_users_per_sec", ("ml100k", "cpu_baseline", "eval", "value")),
                        ("ml100k_eval_ndcg10", ("ml100k", "cpu_baseline", "eval", "ndcg@10"))):
            v = _get(line, *path)
            if v is not None:
                c[k] = v
        for leg, unit in (("ngcf", "triplets_per_sec"), ("multivae", "users_per_sec")):
            v = _get(line, leg, "cpu_baseline", "value")
            if v is not None:
                c["%s_%s" % (leg, unit)] = v
        out["cpu_baseline"] = c
    else:
        out["cpu_baseline"] = None
    out["full_line"] = "bench_full.json next to bench.py / in gpurun_out (committed per round as profiles/rNN_bench.json)"
    return out
