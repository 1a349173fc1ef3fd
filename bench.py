#!/usr/bin/env python
"""bench.py — BPR triplets/sec (+ eval users/sec, NDCG@10) on LightGCN-gowalla, MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of B=1024 BPR triplets of the
gowalla-shaped synthetic graph (BASELINE.json configs[2]): device-side sampling of the
epoch stream (amortised: one launch per epoch, inside the timed region), LightGCN
propagation forward (3 SpMM), BPR head, propagation backward (3 SpMM), dense TF-Adam.
All inputs are resident in HBM before the timed region.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 matrix peak (same guide); v_mfma_f32_32x32x2_f32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--shape", default="gowalla",
                    help="gowalla | ml-100k (host-generated twins) | config4 (BASELINE configs[3], generated "
                         "on the device; needs --dp-mode rowshard; --scale shrinks it)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=1024)     # conf/LightGCN.properties:5
    ap.add_argument("--layers", type=int, default=3)       # BASELINE.json configs[2]
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--eval-batch", type=int, default=32768)
    ap.add_argument("--dp-mode", choices=("replicated", "triplets", "allreduce", "rowshard"), default="replicated",
                    help="N>1: every rank generates the epoch stream of the GLOBAL batch itself (counter-based "
                         "sampler: same seed, same stream — no exchange at all; default), all-gather the "
                         "batch ids, all-reduce dL/dE0, or row-sharded tables (all-gather per hop + "
                         "all-to-all lookups; config 4 path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=24)
    ap.add_argument("--eval-mode", choices=("pruned", "materialised"), default="pruned")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-mf", action="store_true")
    return ap.parse_args()


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


def main():
    args = parse()
    import torch
    from neurec_amd import engine as E, parallel, synth
    from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, LightGCNEngine

    comm = parallel.init_from_env()
    if args.gpus != comm.world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, comm.world, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", torch.cuda.current_device())

    # ---------------- workload, resident in HBM before anything is timed
    exchange = comm.active and args.dp_mode == "triplets"
    # The sampler is a counter-based generator (seed, epoch, position): every rank can produce the
    # whole epoch stream for 47 us, so the global batch of world x B triplets needs no id exchange —
    # and, existing a whole epoch ahead, it gets its batch plans from the sampler like a single GPU's.
    replicated = comm.active and args.dp_mode == "replicated"
    global_batch = args.batch * (comm.world if (exchange or replicated) else 1)
    rowshard = args.dp_mode == "rowshard"
    config4 = args.shape == "config4"
    if config4:
        # BASELINE configs[3] (U = 10^7, I = 10^6, E = 2·10^8 at --scale 1): the graph is generated
        # ON THE DEVICE (Philox counter stream, SURVEY 8d: never materialised on the host), every rank
        # builds only its own row block of the adjacency, tables are row-sharded.
        if not rowshard:
            raise SystemExit("--shape config4 runs with --dp-mode rowshard (row-sharded tables)")
        from neurec_amd import parallel as par
        from neurec_amd.sharded import ShardedLightGCN
        U, I, n_edges = (max(int(x * args.scale), 64) for x in synth.CONFIG4)
        tr_ptr, tr_idx = synth.device_interactions(U, I, n_edges, seed=2018, device=dev)
        n_train = int(tr_ptr[-1])
        part = par.BipartitePartition(U, I, comm.world)     # every rank: a slice of the users AND of the items
        ur, ir = part.users_of(comm.rank), part.items_of(comm.rank)
        rows = synth.device_lightgcn_rank_rows(tr_ptr, tr_idx, U, I, ur, ir)
        lim = float(np.sqrt(6.0 / (U + I + args.dim)))
        g = torch.Generator(device=dev); g.manual_seed(2017 + comm.rank)
        E0 = (torch.rand((ur[1] - ur[0]) + (ir[1] - ir[0]), args.dim, generator=g, device=dev) * 2 - 1) * lim
        lg = ShardedLightGCN(comm, None, U, I, E0, args.layers, 0.01, 1e-3, args.batch, local_rows=rows)
        del rows
        trc, tec, train, test = E.DeviceCSR(tr_ptr, tr_idx, I), None, None, None
        args.no_eval, args.no_mf, args.no_cpu_baseline = True, True, True
        train_nnz = n_train
    else:
        train, test = synth.interactions(args.shape, seed=2018, scale=args.scale)
        U, I = train.shape
        train_nnz = train.nnz
        coo = train.tocoo()
        from neurec_amd.graph import lightgcn_adjacency
        A = lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
        E0 = synth.xavier_uniform(U + I, args.dim, np.random.RandomState(2017))
        if rowshard:
            from neurec_amd.sharded import ShardedLightGCN
            lg = ShardedLightGCN(comm, A, U, I, E0, args.layers, 0.01, 1e-3, args.batch)
        else:
            lg = LightGCNEngine(A, U, I, E0, args.layers, 0.01, 1e-3,        # lr, reg: conf/LightGCN.properties
                                global_batch)
        trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
    # single GPU / all-reduce / row-shard modes step on the sampler's own batches: their batch plans
    # (the order of the duplicate-row gradient sums) are sorted once per epoch by the sampler; in
    # the id-exchange mode the global batch only exists after the all-gather: sorted inside the step
    sampler = BprEpochSampler(trc, I, neg_num=1, batch_size=global_batch if replicated else args.batch,
                              shuffle=True, seed=2018,
                              rank=0 if replicated else comm.rank, world=1 if replicated else comm.world,
                              plan_users=None if (exchange or rowshard) else U)
    loss2 = torch.zeros(2, device=dev)
    grad_sync = comm.allreduce_sum_ if (comm.active and not exchange and not rowshard and not replicated) \
        else None

    def batch_stream():
        while True:
            for k, b in enumerate(sampler.batches()):
                if k == 0 and rowshard:
                    # routing counts of the whole epoch: one pass, one device->host copy — the
                    # steps then run without host synchronisation (sharded.RowRouter.plan_epoch)
                    lg.plan_epoch(sampler._users[:sampler.n_local], sampler._pos[:sampler.n_local],
                                  sampler._neg[:sampler.n_local], args.batch)
                b.index = k
                if b[0].numel() == sampler.batch_size:  # fixed-size steps for the timed region
                    yield b
    stream = batch_stream()
    inflight = [comm.allgather_cat_start(next(stream))] if exchange else None

    def run_steps(n, loss_out=None):
        # the reference never fetches LightGCN's loss while training (LightGCN.py:173-180), so
        # the timed steps do not reduce it either; it is evaluated once after the timed region
        for _ in range(n):
            if exchange:
                # ids of this step were all-gathered while the previous step ran; start the
                # next step's gather before launching this step
                token = inflight[0]
                inflight[0] = comm.allgather_cat_start(next(stream))
                bu, bp, bn = comm.allgather_cat_finish(token)
                lg.step(bu, bp, bn, loss_out)
            elif rowshard:
                b = next(stream)
                lg.step(b[0], b[1], b[2], loss_out, batch_index=b.index)
            else:
                b = next(stream)
                lg.step(b[0], b[1], b[2], loss_out, grad_sync=grad_sync, plan=b.plan)

    run_steps(args.warmup)
    torch.cuda.synchronize(); comm.barrier()
    epochs_before = sampler.epoch
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize(); comm.barrier()
    dt = comm.max_float(time.perf_counter() - t0)
    # what the timed region held: the sampler (+ batch-plan) launch happens once per epoch of
    # len(sampler) steps, so a short run may contain none — said here rather than implied
    timed_region = {"steps": args.steps, "steps_per_epoch": len(sampler),
                    "sampler_launches": sampler.epoch - epochs_before,
                    "batch_plan_launches": (sampler.epoch - epochs_before) if sampler.plans else
                    ("one per step (sorted inside the step)" if not rowshard else 0)}
    triplets_per_s = comm.world * args.steps * args.batch / dt
    run_steps(1, loss2)                                  # untimed: loss of one more step, for the record
    if exchange:
        comm.allgather_cat_finish(inflight[0])           # drain the prefetched id gather

    # ---------------- roofline of the dominant kernel (CSR SpMM): HIP events on the launch stream
    reps = 20 if lg.A.nnz < 50_000_000 else 3
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lg.propagate(); torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        if rowshard:                                     # local SpMM launches only (no collectives)
            for k in range(2 * args.layers):
                lg.A.matmul(lg.X, out=(lg.Ya, lg.Yb)[k % 2], addend=lg.H)
            continue
        # the plain pass Y = A·X — the launch the step's full hops are (no running-sum or addend streams:
        # the 49.6 MB of SURVEY 8d's formula are exactly its bytes), L forward + L backward operands
        src = lg.E0
        for k in range(args.layers):
            lg.A.matmul(src, out=(lg.Ea, lg.Eb)[k % 2])
            src = (lg.Ea, lg.Eb)[k % 2]
        g = lg.H
        for k in range(args.layers):
            lg.At.matmul(g, out=(lg.Ga, lg.Gb)[k % 2])
            g = (lg.Ga, lg.Gb)[k % 2]
    ev1.record(); torch.cuda.synchronize()
    spmm_ms = ev0.elapsed_time(ev1) / (reps * 2 * max(args.layers, 1))
    spmm_bytes = lg.A.algorithmic_bytes(args.dim)
    achieved = spmm_bytes / (spmm_ms * 1e-3) / 1e9
    kernel = lg.A.full_pass_kernel(args.dim)
    # traffic: PMC counters cannot be read inside this process; the committed rocprofv3 --pmc passes
    # over this same command (scripts/gpu_pmc.sh -> profiles/r02_pmc_traffic.json) are reported when
    # they are for the kernel that ran, the default workload AND the SpMM sources they were measured
    # on (hash stamped in the file) — otherwise null
    traffic, traffic_note = None, "no PMC pass for this kernel / workload"
    here = os.path.dirname(os.path.abspath(__file__))
    pmc_file = os.path.join(here, "profiles", "r02_pmc_traffic.json")
    default_workload = (args.shape, args.scale, args.dim, args.layers) == ("gowalla", 1.0, 64, 3)
    if default_workload and os.path.isfile(pmc_file):
        import hashlib
        h = hashlib.sha256()
        for name in ("spmm_blocked.hip", "spmm.hip"):
            with open(os.path.join(here, "neurec_amd", "csrc", name), "rb") as fh:
                h.update(fh.read())
        with open(pmc_file) as fh:
            doc = json.load(fh)
        if doc.get("_spmm_sources_sha16") != h.hexdigest()[:16]:
            traffic_note = "stale: csrc/spmm*.hip changed since the PMC pass of profiles/r02_pmc_traffic.json"
        else:
            hit = [v for k, v in doc["kernels"].items() if k.replace(" ", "") == kernel.replace(" ", "")]
            if hit:
                traffic = hit[0]["traffic_bytes_per_launch"]
                traffic_note = ("bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc "
                                "passes (profiles/r02_pmc_traffic.json, sources %s)" % doc["_spmm_sources_sha16"])
    roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_note": traffic_note, "bytes_per_launch": spmm_bytes,
                "us_per_launch": spmm_ms * 1e3, "launches_per_step": 2 * args.layers,
                "step_algorithmic_bytes": lg.step_bytes() if hasattr(lg, "step_bytes") else None,
                "step_bytes_survey_8d": lg.step_bytes_survey() if hasattr(lg, "step_bytes_survey") else None}
    if roofline["step_algorithmic_bytes"]:
        # the whole step against the same roof: every launch's algorithmic bytes / the step time
        roofline["step_frac"] = roofline["step_algorithmic_bytes"] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS
        roofline["step_frac_survey_8d"] = roofline["step_bytes_survey_8d"] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS

    # ---------------- BPR-MF on the same interactions (BASELINE configs[1]: d=64, B=512) — reported
    # next to the headline, not instead of it.  A step = fused gather/BPR/scatter kernel + the two
    # TF-sparse Adam sweeps (every row of both tables decays each step).
    mf_info = None
    if comm.rank == 0 and not args.no_mf:
        from neurec_amd.trainer import MFEngine
        rs = np.random.RandomState(2017)
        mf = MFEngine((rs.randn(U, 64) * 0.01).astype(np.float32), (rs.randn(I, 64) * 0.01).astype(np.float32),
                      0.001, 0.0, 512)                                   # conf/MF.properties
        mf_sampler = BprEpochSampler(trc, I, neg_num=1, batch_size=512, shuffle=True, seed=2018,
                                     plan_users=U)
        # the batch loop of MF.train_model runs natively (MFEngine.run_batches -> nrhip_mf_steps): a
        # Python loop enqueues ~12 us per step, about what the one-launch step takes on the GPU
        mu, mp, mn, mplans = mf_sampler.epoch_stream()
        avail = mu.numel() // 512
        w_steps = min(50, avail // 4)
        t_steps = max(min(400, avail - w_steps), 1)
        n_warm, n_timed = w_steps * 512, t_steps * 512
        mf_loss = torch.zeros(max(t_steps, w_steps, 1), 2, device=dev)
        cut = lambda lo, hi: (mu[lo:hi], mp[lo:hi], mn[lo:hi])
        mf.run_batches(*cut(0, n_warm), 512, mf_loss, mplans[:3 * n_warm])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mf.run_batches(*cut(n_warm, n_warm + n_timed), 512, mf_loss, mplans[3 * n_warm:3 * (n_warm + n_timed)])
        torch.cuda.synchronize()
        mf_dt = (time.perf_counter() - t0) / t_steps
        # the same steps with TF's literal all-rows sweep (the checker) for the record
        mf_sweep = MFEngine((rs.randn(U, 64) * 0.01).astype(np.float32), (rs.randn(I, 64) * 0.01).astype(np.float32),
                            0.001, 0.0, 512, lazy=False)
        mf_sweep.run_batches(*cut(0, n_warm), 512, mf_loss, mplans[:3 * n_warm])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mf_sweep.run_batches(*cut(n_warm, n_warm + n_timed), 512, mf_loss, mplans[3 * n_warm:3 * (n_warm + n_timed)])
        torch.cuda.synchronize()
        sweep_dt = (time.perf_counter() - t0) / t_steps
        touched = 512 * (72 * 64 + 12)           # 3 rows x (read + write of p, m, v) + ids, per triplet
        mf_info = {"triplets_per_sec": 512 / mf_dt, "ms_per_step": mf_dt * 1e3, "batch": 512, "dim": 64,
                   "optimizer": "TF-1.12 sparse Adam by exact lazy replay (bit-identical to the all-rows sweep), "
                                "gradient + optimiser in one launch on double-buffered tables",
                   "roofline": {"bound": "hbm", "bytes_per_step": touched, "unit": "GB/s",
                                "achieved": touched / mf_dt / 1e9, "peak": HBM_PEAK_GBS,
                                "frac": touched / mf_dt / 1e9 / HBM_PEAK_GBS,
                                "note": "SURVEY 8d bound (72 d + 12) B per triplet; one launch whose critical path is a "
                                        "chain of ~5 dependent memory round trips (plan key -> ids -> stamps + rows -> "
                                        "ordered row sums -> Adam -> store -> loss reduction): latency-bound, not "
                                        "bandwidth-bound"},
                   "sweep_ms_per_step": sweep_dt * 1e3,
                   "sweep_GBps": 2 * 4 * (U + I) * 64 * 4 / sweep_dt / 1e9}
        if not args.no_eval:
            # configs[1] names the evaluator too: the BPR-MF tables through the same full-rank path
            tu = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).to(dev)
            mf_ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=args.eval_batch,
                                      pruned=args.eval_mode == "pruned")
            mf_ev.evaluate_factors(mf.P, mf.Q, tu)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mm = mf_ev.evaluate_factors(mf.P, mf.Q, tu)
            torch.cuda.synchronize()
            mf_edt = time.perf_counter() - t0
            mf_info["eval"] = {"users_per_sec": tu.numel() / mf_edt, "ms": mf_edt * 1e3, "n_users": int(tu.numel()),
                               "ndcg@10": float(mm[2 * 20 + 9])}
            del mf_ev

    # ---------------- evaluation leg: users/sec + NDCG@10 (full rank, all users with test items)
    eval_info = None
    if not args.no_eval:
        test_users = np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)
        mine = torch.from_numpy(parallel.shard_users(test_users, comm.rank, comm.world)).to(dev)
        ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=args.eval_batch,
                               pruned=args.eval_mode == "pruned")

        # replicas that ran the same global steps hold bit-identical tables (row gradients are
        # summed in batch order, nothing is unordered): no re-alignment before sharded scoring

        def evaluate():
            eu, ei = lg.final_embeddings()
            sums = ev.evaluate_factors(eu.contiguous(), ei.contiguous(), mine) * mine.numel()
            if not comm.active:                      # one rank: the sums are the totals (no round trip)
                return np.asarray(sums, np.float64) / len(test_users)
            t = torch.from_numpy(np.asarray(sums, np.float64)).to(dev)
            comm.allreduce_sum_(t)
            return (t / len(test_users)).cpu().numpy()
        evaluate()
        torch.cuda.synchronize(); comm.barrier()
        t0 = time.perf_counter()
        means = evaluate()
        torch.cuda.synchronize(); comm.barrier()
        dte = comm.max_float(time.perf_counter() - t0)
        eval_info = {"users_per_sec": len(test_users) / dte, "ms": dte * 1e3,
                     "n_users": int(len(test_users)), "ndcg@10": float(means[2 * 20 + 9]),
                     "recall@20": float(means[1 * 20 + 19]),
                     "design": ("pruned: tile maxima in the fp32-MFMA scoring loop (no score matrix) -> top-21 "
                                "32-item tiles per user rescored + ranked; tie rows redone from full rows"
                                if args.eval_mode == "pruned" else
                                "materialised scores: fp32-MFMA GEMM -> HBM -> select kernel; scoring of "
                                "batch b+1 overlaps ranking of batch b (two streams, two slabs)"),
                     "rows_redone_for_ties": getattr(ev, "n_flagged", 0) if args.eval_mode == "pruned" else None}

        # rooflines of the evaluation's two halves, HIP events on the launch stream around the kernels
        # of the first batch (north_star: MFMA for the scoring matmul, HBM GB/s for the top-K)
        if args.eval_mode == "pruned" and mine.numel() > 0:
            eu, ei = lg.final_embeddings()
            eu, ei = eu.contiguous(), ei.contiguous()
            ub = mine[:args.eval_batch]
            nb, d_e, top_k = ub.numel(), eu.shape[1], 20
            ev._gemm.prepare(ei)
            M = ev._gemm.tile_maxima(eu, ub, trc)
            per = torch.empty((nb, 5 * top_k), dtype=torch.float32, device=dev)
            flg = torch.zeros(nb, dtype=torch.int32, device=dev)
            E.eval_tiles(M, eu, ev._gemm, ub, trc, tec, [1, 2, 4, 3, 5], top_k, per, flg)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                M = ev._gemm.tile_maxima(eu, ub, trc)
            e1.record()
            for _ in range(3):
                E.eval_tiles(M, eu, ev._gemm, ub, trc, tec, [1, 2, 4, 3, 5], top_k, per, flg)
            e2.record()
            torch.cuda.synchronize()
            t_score, t_rank = e0.elapsed_time(e1) / 3e3, e1.elapsed_time(e2) / 3e3       # seconds
            flops = 2.0 * I * d_e * nb
            tiles = 2 * ((I + 63) // 64)
            # level 2, algorithmic HBM bytes: per user its tile maxima and factor row read and M*K metrics
            # written; the k-major item copy and the train/test lists read once.  (The top_k+1 rescored
            # tiles per user are gathers from that L2-resident item copy: reported apart, not HBM bytes.)
            rank_bytes = nb * (tiles * 4 + d_e * 4 + 5 * top_k * 4) + I * d_e * 4 + \
                (int(train.indptr[-1]) + int(test.indptr[-1])) * 4
            rescore_l2_bytes = nb * (top_k + 1) * 64 * d_e * 4
            eval_info["roofline"] = {
                "bound": "mfma", "kernel": "score_tilemax_kernel<32>", "users": nb,
                "achieved": flops / t_score / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / t_score / 1e12 / MFMA_F32_PEAK_TFLOPS, "ms": t_score * 1e3,
                "flops_per_user": 2.0 * I * d_e}
            eval_info["roofline_topk"] = {
                "bound": "hbm", "kernel": "rescore_tiles_kernel + select_rows_kernel + metrics_kernel (nrhip_eval_tiles)",
                "achieved": rank_bytes / t_rank / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": rank_bytes / t_rank / 1e9 / HBM_PEAK_GBS, "ms": t_rank * 1e3,
                "bytes": rank_bytes, "rescore_gather_bytes_from_l2": rescore_l2_bytes,
                "rescore_gather_GBps": rescore_l2_bytes / t_rank / 1e9,
                "note": "pruned design: the [users][I] score matrix is never written; the top-K works on "
                        "tile maxima and %d rescored tiles per user gathered from the L2-resident item "
                        "copy, so its HBM bytes are small and the phase is bound by those L2 gathers and "
                        "by latency, not by HBM" % (top_k + 1)}

    line = {
        "metric": "BPR triplets/sec (LightGCN-%s)" % args.shape, "value": triplets_per_s,
        "unit": "triplets/s", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "data_note": "train AND test interactions are synthetic twins of the named shape (neurec_amd/synth.py); SURVEY "
                     "8d names the reference's real dataset/gowalla.test as the test split, but the reference tree does "
                     "not exist on the GPU box and gowalla.train is absent from the reference tree altogether",
        "config": {"workload": "LightGCN on synthetic %s-shaped interactions (U=%d, I=%d, E=%d), "
                               "%d layers, dim %d, B=%d per GPU, adj=pre, Adam lr=0.01 reg=1e-3"
                               % (args.shape, U, I, train_nnz, args.layers, args.dim, args.batch),
                   "global_batch": comm.world * args.batch,
                   "parallelism": ("dp%d (replicated tables; every rank generates the same global epoch "
                                   "stream from the shared seed and steps on the global batch of "
                                   "%d: no exchange in training, tables bit-identical on all ranks)"
                                   % (comm.world, global_batch) if replicated else
                                   "dp%d (replicated tables; per step one all-gather of 12 B/triplet "
                                   "of ids, every rank steps on the global batch)" % comm.world
                                   if exchange else
                                   "rowshard%d (tables row-sharded; all-gather per hop, all-to-all "
                                   "row lookups, owner-local Adam)" % comm.world if rowshard else
                                   "dp%d (replicated tables, one all-reduce of dL/dE0 per step)" % comm.world)
                   if comm.active else "single GPU"},
        "final_loss": [float(x) for x in loss2.cpu().numpy()], "timed_region": timed_region,
        "eval": eval_info, "mf": mf_info, "roofline": roofline,
        "device": E.device_info(),
    }
    if comm.rank == 0 and comm.world == 1 and not args.no_cpu_baseline:
        tables = None
        if eval_info is not None:
            eu, ei = lg.final_embeddings()
            tables = (eu.cpu().numpy(), ei.cpu().numpy())
        cb = cpu_baseline(train, test, E0, args, eval_tables=tables)
        sample_users = cb["eval"].pop("users")
        if eval_info is not None:
            # same embeddings, same users: the GPU evaluator's NDCG@10 next to the reference C++ fed
            # by np.matmul (north_star: equal within 1e-5; BLAS and the fmaf chain differ in the last
            # ulps of a score, so a near-tie may swap)
            mine = ev.evaluate_factors(eu.contiguous(), ei.contiguous(),
                                       torch.from_numpy(sample_users).to(dev), exact_mean=True)
            eval_info["ndcg10_oracle_absdiff"] = abs(float(mine[2 * 20 + 9]) - cb["eval"]["ndcg@10"])
            eval_info["ndcg10_oracle_sample"] = "%d users, np.matmul scores + %s C++ evaluator" % (
                len(sample_users), "the reference's own" if cb["eval"]["kind"] == "reference" else "the oracle's")
        line["cpu_baseline"] = cb
    else:
        line["cpu_baseline"] = None
    if comm.rank == 0:
        print(json.dumps(line))
    comm.shutdown()


if __name__ == "__main__":
    main()
