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

from bench_legs import (HBM_PEAK_GBS, L2_PEAK_GBS, MFMA_F32_PEAK_TFLOPS, MFMA_BF16_PEAK_TFLOPS, MFMA_I8_PEAK_TOPS, _hip_timed, _cpu_timed, _median_wall, leg_mf, leg_ngcf, leg_multivae, leg_config4, _config4_eval, leg_config4_partitions, cpu_baseline, _get, compact_line)  # noqa: E402,F401  (the legs live in bench_legs.py)


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
    ap.add_argument("--dp-mode", choices=("colshard", "replicated", "triplets", "allreduce", "rowshard"), default=None,
                    help="N>1 (default: colshard — every rank holds dim/N COLUMNS of the table for all nodes and steps "
                         "on the whole global batch: the propagation, the gradient rows and Adam are column-wise, so the "
                         "step's one exchange is an RCCL all-gather of the per-triplet partial inner products, 12 B per "
                         "triplet and rank; nothing is computed twice.  allreduce: every rank back-propagates ITS B "
                         "triplets, RCCL sums dL/dE0; the batch-independent full hops are repeated on every rank and "
                         "labelled so).  rowshard: tables "
                         "row-sharded, nothing repeated (all-gather per hop + all-to-all lookups; the config-4 "
                         "path).  Opt-in, fully redundant compute: replicated (every rank generates the global "
                         "batch itself and runs the whole step on it, no exchange) and triplets (ids all-gathered)")
    ap.add_argument("--rowshard-hop", choices=("sliced", "allgather", "chunked", "reduce"), default=None,
                    help="--dp-mode rowshard: how a propagation hop gets the other ranks' rows (neurec_amd/sharded.py; "
                         "default: sliced — column slabs all-gathered under the one-launch SpMM of the previous slab, "
                         "exact; reduce — item rows as per-rank partials + all-to-all, 5x fewer bytes at config 4, "
                         "within fp32 rounding)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the full line (every leg with kernels, notes and samples) instead of the compact one the "
                         "driver parses; the full line always goes to bench_full.json as well")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=24)
    ap.add_argument("--eval-mode", choices=("pruned", "materialised"), default="pruned")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-mf", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="skip the NGCF / Mult-VAE legs (BASELINE configs[4])")
    ap.add_argument("--no-config4", action="store_true", help="skip the config-4 slice leg (BASELINE configs[3])")
    ap.add_argument("--config4-eval-batch", type=int, default=65536,
                    help="users per batch of the config-4 evaluation leg (tile maxima: 4 B x 31,250 tiles per user = 8 GB "
                         "at 65,536; measured 2.04 / 2.19 / 2.31 / 2.38 M users/s at 8,192 / 16,384 / 32,768 / 65,536: "
                         "the planned strikes and the tile buckets of level 2 fill their 32-pair chunks only when a "
                         "batch brings >= 32 pairs per tile)")
    ap.add_argument("--config4-scale", type=float, default=1.0,
                    help="fraction of BASELINE configs[3] (10^7 users, 10^6 items, 2*10^8 interactions) the one-GPU "
                         "leg runs at on this one GPU; 1.0 is the full size (213 ms/step)")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` from a plain shell (no launcher, WORLD_SIZE unset): start the N ranks here — one
    process per GPU, the environment torch.distributed.run would set (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE),
    a file-store rendezvous instead of a port, the same command line — and wait for them.  Rank 0's stdout (the ONE JSON
    line) is this process's stdout.  If a rank fails the others are stopped (exactly the PIDs started here) and its
    exit code is returned."""
    import subprocess
    import tempfile
    # rendezvous through a file store (parallel._rendezvous): no port is picked here and lost before the ranks bind it
    store = os.path.join(tempfile.mkdtemp(prefix="neurec_bench_"), "store")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", NEUREC_DIST_INIT_FILE=store, NEUREC_BENCH_SELF_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch
    from neurec_amd import engine as E, parallel, synth
    from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, LightGCNEngine

    comm = parallel.init_from_env()
    no_eval_legs = args.no_eval
    if args.dp_mode is None:
        # one rank: the modes coincide.  N ranks: the column-sharded engine when the width divides
        args.dp_mode = "replicated" if not comm.active else ("colshard" if args.dim % comm.world == 0 else "allreduce")
    if args.gpus != comm.world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's --nproc-per-node and --gpus must agree "
                         "(a plain `python bench.py --gpus %d` starts its own ranks)"
                         % (args.gpus, comm.world, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", torch.cuda.current_device())

    # ---------------- workload, resident in HBM before anything is timed
    exchange = comm.active and args.dp_mode == "triplets"
    # The sampler is a counter-based generator (seed, epoch, position): every rank can produce the
    # whole epoch stream for 47 us, so the global batch of world x B triplets needs no id exchange —
    # and, existing a whole epoch ahead, it gets its batch plans from the sampler like a single GPU's.
    colshard = comm.active and args.dp_mode == "colshard"
    # (column-sharded ranks step on the same global batch, produced locally like the replicated mode's)
    replicated = comm.active and args.dp_mode in ("replicated", "colshard")
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
        lg = ShardedLightGCN(comm, None, U, I, E0, args.layers, 0.01, 1e-3, args.batch, local_rows=rows,
                             hop=args.rowshard_hop)
        del rows
        trc, tec, train, test = E.DeviceCSR(tr_ptr, tr_idx, I), None, None, None
        config4_eval = not args.no_eval                      # its own evaluation leg (_config4_eval), after the timed steps
        args.no_eval, args.no_mf, args.no_cpu_baseline = True, True, True
        train_nnz = n_train
    else:
        real_test = os.path.join(ROOT, "tests", "golden", "gowalla_test_split.npz")
        data_test = "synthetic"
        if args.shape == "gowalla" and args.scale == 1.0 and os.path.isfile(real_test):
            # the test split is the reference's real dataset/gowalla.test (committed as a fixture —
            # the reference tree does not travel); gowalla.train is absent from the reference tree, so the
            # train side is drawn around that split (E = 810,128, the LightGCN-paper split's size)
            train, test = synth.interactions_around_test(synth.load_test_split(real_test), 810128, seed=2018)
            data_test = "the reference's dataset/gowalla.test (217,242 pairs)"
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
            lg = ShardedLightGCN(comm, A, U, I, E0, args.layers, 0.01, 1e-3, args.batch, hop=args.rowshard_hop)
        elif colshard:
            from neurec_amd.colshard import ColumnShardedLightGCN
            lg = ColumnShardedLightGCN(comm, A, U, I, E0, args.layers, 0.01, 1e-3, global_batch)
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
            elif colshard:
                b = next(stream)
                lg.step(b[0], b[1], b[2], loss_out, plan=b.plan)
            else:
                b = next(stream)
                lg.step(b[0], b[1], b[2], loss_out, grad_sync=grad_sync, plan=b.plan)

    # The host-side setup above (graph, adjacency, schedules: seconds) left the device idle and clocked down; a short
    # run (the driver's --steps 20 --warmup 5 is 5 ms of device work) would time the clock ramp, not the step: 0.210 ms
    # per step against 0.199 after 300 warmup steps (profiles/r05_exp_short_window.txt).  Untimed, state-free device
    # work first — plain propagation passes into a scratch buffer.
    # (the PLAIN pass Y = A·X into a scratch layer buffer — the launch the step's full hops are, so the per-kernel
    #  averages of a rocprofv3 / PMC pass over this command stay those of that launch; no collective inside)
    eng0 = lg.local if colshard else lg
    spin = (lambda: eng0.local_pass(0)) if rowshard else (lambda: eng0.A.matmul(eng0.E0, out=eng0.Ea))
    n_spin = max(2, min(600, int(9e8 / max(2 * int(train_nnz), 1))))
    t_spin = time.perf_counter()
    for _ in range(n_spin):
        spin()
    torch.cuda.synchronize()
    spin_ms = (time.perf_counter() - t_spin) * 1e3
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
                    "device_spin_up_before_warmup": {"plain_passes": n_spin, "ms": spin_ms,
                                                     "note": "untimed, state-free (scratch buffers): brings the device to "
                                                             "its running clocks after seconds of host-side setup"},
                    "sampler_launches": sampler.epoch - epochs_before,
                    "batch_plan_launches": (sampler.epoch - epochs_before) if sampler.plans else
                    ("one per step (sorted inside the step)" if not rowshard else 0)}
    triplets_per_s = comm.world * args.steps * args.batch / dt
    run_steps(1, loss2)                                  # untimed: loss of one more step, for the record
    # ONE WHOLE EPOCH, timed (the metric is E / epoch wall time, sampler included): the
    # sampler's launch, the batch plans and EVERY batch of the permuted stream, the short last one too
    # (LightGCN.py:168-180 with data_iterator.py's drop_last=False) — whatever --steps is.  Where the ranks step in
    # lock-step on their OWN slices (all-reduce / row-shard with N > 1: a collective per step, a rank's slice may hold
    # one batch more than another's) the epoch is its full batches.  Skipped when an epoch would take more than ~5 s
    # (config 4 at full size: 24 k steps of 0.14 s).
    epoch_timed = None
    n_full = int(comm.max_float(float(len(sampler))))
    if not exchange and n_full * dt / args.steps < 5.0:
        lockstep = comm.active and not replicated
        torch.cuda.synchronize(); comm.barrier()
        e_before = sampler.epoch
        t0 = time.perf_counter()
        n_tr, n_st = 0, 0
        if lockstep:
            n_st = int(-comm.max_float(-float(sampler.n_local // sampler.batch_size)))     # full batches every rank has
            run_steps(n_st)
            n_tr = comm.world * n_st * args.batch
        else:
            for k, b in enumerate(sampler.batches()):
                if k == 0 and rowshard:
                    lg.plan_epoch(sampler._users[:sampler.n_local], sampler._pos[:sampler.n_local],
                                  sampler._neg[:sampler.n_local], args.batch)
                if rowshard:
                    lg.step(b[0], b[1], b[2], None, batch_index=k)
                elif colshard:
                    lg.step(b[0], b[1], b[2], None, plan=b.plan)
                else:
                    lg.step(b[0], b[1], b[2], None, grad_sync=grad_sync, plan=b.plan)
                n_tr += b[0].numel()
                n_st += 1
        torch.cuda.synchronize(); comm.barrier()
        dt_e = comm.max_float(time.perf_counter() - t0)
        epoch_timed = {"value": n_tr / dt_e, "unit": "triplets/s", "ms": dt_e * 1e3, "steps": n_st, "triplets": n_tr,
                       "sampler_launches": sampler.epoch - e_before, "short_last_batch": (not lockstep) and n_tr % (
                           sampler.batch_size) != 0,
                       "vs_steps_window": (n_tr / dt_e) / triplets_per_s,
                       "agrees_with_steps_window_within_2pct": abs((n_tr / dt_e) / triplets_per_s - 1.0) <= 0.02}
    # SURVEY 8d defines the metric "sampler included" = E / epoch wall time: the per-epoch launches (sampler,
    # batch plans) measured on their own and charged to an epoch of len(sampler) steps at the measured step time
    epoch_ms = _hip_timed(sampler.sample_epoch, 3, 1)
    # (the same number on every rank: a rank's slice of the stream may hold one batch more or less than another's, and the
    #  untimed training before the eval leg below runs `f(steps_per_epoch)` steps with collectives inside)
    steps_per_epoch = int(comm.max_float(float(len(sampler))))
    n_epoch = sampler.n_local if not replicated else sampler.n_local // max(comm.world, 1)
    timed_region["epoch_launch_ms"] = epoch_ms
    epoch_amortised = comm.world * n_epoch / (steps_per_epoch * dt / args.steps + epoch_ms * 1e-3)
    # the evaluation leg wants a MODEL, not 25 steps from Xavier noise: train on, untimed, until two
    # epochs of the stream have been stepped (0.35 s at gowalla) — the NDCG@10 of the line and its comparison with the
    # reference's evaluator are then numbers about ranking
    eval_pretrain_steps = 0
    if not args.no_eval and not config4:
        eval_pretrain_steps = max(0, 2 * steps_per_epoch - (args.warmup + args.steps + 1))
        t_pre = time.perf_counter()
        run_steps(eval_pretrain_steps)
        torch.cuda.synchronize()
        timed_region["untimed_training_before_the_eval_leg"] = {
            "steps": eval_pretrain_steps, "seconds": time.perf_counter() - t_pre,
            "total_steps_trained": eval_pretrain_steps + args.warmup + args.steps + 1}
    if exchange:
        comm.allgather_cat_finish(inflight[0])           # drain the prefetched id gather

    # ---------------- roofline of the dominant kernel (CSR SpMM): HIP events on the launch stream
    full = lg                                            # what evaluation asks for the tables
    if colshard:
        lg = lg.local                                    # the rank's own engine: its kernels are what is profiled below
    wdim = lg.d if colshard else (lg.w if rowshard else args.dim)   # width the rank's kernels run at (its columns / slab)
    reps = 20 if lg.A.nnz < 50_000_000 else 3
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lg.propagate(); torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        if rowshard:                                     # local SpMM launches only (no collectives)
            for k in range(2 * args.layers):
                lg.local_pass(k)
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
    spmm_ms = ev0.elapsed_time(ev1) / (reps * 2 * max(args.layers, 1))     # row-sharded: per hop (all its slab launches)
    spmm_bytes = lg.A.algorithmic_bytes(args.dim if rowshard else wdim)
    achieved = spmm_bytes / (spmm_ms * 1e-3) / 1e9
    kernel = lg.A.full_pass_kernel(wdim)
    # traffic: PMC counters cannot be read inside this process; the committed rocprofv3 --pmc passes
    # over this same command (scripts/gpu_pmc.sh -> profiles/r02_pmc_traffic.json) are reported when
    # they are for the kernel that ran, the default workload AND the SpMM sources they were measured
    # on (hash stamped in the file) — otherwise null
    traffic, traffic_note = None, "no PMC pass for this kernel / workload"
    here = os.path.dirname(os.path.abspath(__file__))
    pmc_file = next((f for f in (os.path.join(here, "profiles", "r0%d_pmc_traffic.json" % r) for r in (6, 5, 4, 3, 2))
                     if os.path.isfile(f)), "")
    default_workload = (args.shape, args.scale, args.dim, args.layers) == ("gowalla", 1.0, 64, 3)
    if default_workload and os.path.isfile(pmc_file) and not colshard:
        import hashlib
        h = hashlib.sha256()
        for name in ("spmm_blocked.hip", "spmm.hip"):
            with open(os.path.join(here, "neurec_amd", "csrc", name), "rb") as fh:
                h.update(fh.read())
        with open(pmc_file) as fh:
            doc = json.load(fh)
        if doc.get("_spmm_sources_sha16") != h.hexdigest()[:16]:
            traffic_note = "stale: csrc/spmm*.hip changed since the PMC pass of profiles/%s" % os.path.basename(pmc_file)
        else:
            hit = [v for k, v in doc["kernels"].items() if k.replace(" ", "") == kernel.replace(" ", "")]
            if hit:
                traffic = hit[0]["traffic_bytes_per_launch"]
                traffic_note = ("bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc "
                                "passes (profiles/%s, sources %s)" % (os.path.basename(pmc_file), doc["_spmm_sources_sha16"]))
    roofline = {"bound": "hbm", "kernel": kernel, "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_note": traffic_note, "bytes_per_launch": spmm_bytes,
                "us_per_launch": spmm_ms * 1e3, "launches_per_step": 2 * args.layers,
                "step_algorithmic_bytes": lg.step_bytes() if hasattr(lg, "step_bytes") else None,
                "step_bytes_survey_8d": lg.step_bytes_survey() if hasattr(lg, "step_bytes_survey") else None}
    # the same kernel's average in the committed rocprofv3 --kernel-trace --stats table of this command (it cannot be
    # collected from inside; VERDICT r3 weak #13: both clocks in the line, not the favourable one)
    for rr in ("r06", "r05", "r04", "r03"):
        stats = os.path.join(here, "profiles", "%s_bench_kernel_stats.csv" % rr)
        if default_workload and os.path.isfile(stats):
            import csv
            with open(stats) as fh:
                hit = [r for r in csv.DictReader(fh) if kernel.replace(" ", "") in r.get("Name", "").replace(" ", "")]
            if hit:
                us = float(hit[0]["AverageNs"]) / 1e3
                roofline["us_per_launch_rocprof"] = us
                roofline["frac_rocprof"] = spmm_bytes / us / 1e3 / HBM_PEAK_GBS
                roofline["rocprof_table"] = "profiles/%s_bench_kernel_stats.csv" % rr
            break
    if roofline["step_algorithmic_bytes"]:
        # the whole step against the same roof: every launch's algorithmic bytes / the step time
        roofline["step_frac"] = roofline["step_algorithmic_bytes"] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS
        roofline["step_frac_survey_8d"] = roofline["step_bytes_survey_8d"] / (dt / args.steps) / 1e9 / HBM_PEAK_GBS

    # ---------------- BPR-MF on the same interactions (BASELINE configs[1]: d=64, B=512) — reported
    # next to the headline, not instead of it (leg_mf)
    mf_info = mf = None
    if comm.rank == 0 and not args.no_mf:
        mf_info, mf = leg_mf(train, test, trc, tec, dev, args.eval_batch, args.eval_mode, not args.no_eval, False)

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
            eu, ei = full.final_embeddings()
            sums = ev.evaluate_factors(eu.contiguous(), ei.contiguous(), mine) * mine.numel()
            if not comm.active:                      # one rank: the sums are the totals (no round trip)
                return np.asarray(sums, np.float64) / len(test_users)
            t = torch.from_numpy(np.asarray(sums, np.float64)).to(dev)
            comm.allreduce_sum_(t)
            return (t / len(test_users)).cpu().numpy()
        evaluate()
        runs = []
        for _ in range(5):                           # wall clock of whole evaluations; the median is reported
            torch.cuda.synchronize(); comm.barrier()
            t0 = time.perf_counter()
            means = evaluate()
            torch.cuda.synchronize(); comm.barrier()
            runs.append(comm.max_float(time.perf_counter() - t0))
        dte = sorted(runs)[len(runs) // 2]
        eval_info = {"users_per_sec": len(test_users) / dte, "ms": dte * 1e3,
                     "ms_runs": [r * 1e3 for r in runs],
                     "n_users": int(len(test_users)), "ndcg@10": float(means[2 * 20 + 9]),
                     "trained_steps": eval_pretrain_steps + args.warmup + args.steps + 1,
                     "recall@20": float(means[1 * 20 + 19]),
                     "search": getattr(ev, "search_used", None),
                     "design": ("pruned: tile maxima from a bounded matrix-core filter (no score matrix; `search` names its arithmetic: "
                                "int8 = 15-bit fixed point in exact integer accumulators, bf16 = three-term bf16 expansion; per-row "
                                "error bound; train strikes as a planned fp32 fix-up pass) -> the best 23 32-item tiles per user "
                                "rescored with the fp32 chain (bucketed by tile, fp32 MFMA) + ranked; each row certified against "
                                "its bound; tie / uncertified rows redone from full fp32 rows"
                                if args.eval_mode == "pruned" else
                                "materialised scores: fp32-MFMA GEMM -> HBM -> select kernel; scoring of "
                                "batch b+1 overlaps ranking of batch b (two streams, two slabs)"),
                     "rows_redone_for_ties": getattr(ev, "n_flagged", 0) if args.eval_mode == "pruned" else None}

        # SURVEY 8d prices the evaluation against the fp32 matrix peak ("fp32 is mandatory"): every RANKED score is the
        # fp32 fmaf chain's, but the tile SEARCH runs on the int8 / bf16 matrix cores, so the whole evaluation can (and does)
        # exceed that roof — said here as a ratio, next to eval_search / eval_rows_redone / eval_fp32_loop_ms
        eval_info["fp32_roof_ratio"] = 2.0 * I * args.dim * len(test_users) / dte / 1e12 / MFMA_F32_PEAK_TFLOPS
        eval_info["fp32_roof_note"] = ("2·I·d flop per user / whole evaluation time / %.1f TFLOP/s fp32-MFMA peak; > 1 is "
                                       "possible because only the ranked scores are fp32 (certified reduced-precision search)"
                                       % MFMA_F32_PEAK_TFLOPS)
        # rooflines of the evaluation's two halves, HIP events on the launch stream around the kernels
        # of the first batch (north_star: MFMA for the scoring matmul, HBM GB/s for the top-K)
        if args.eval_mode == "pruned" and mine.numel() > 0:
            eu, ei = full.final_embeddings()
            eu, ei = eu.contiguous(), ei.contiguous()
            ub = mine[:args.eval_batch]
            nb, d_e, top_k = ub.numel(), eu.shape[1], 20
            ev._gemm.prepare(ei)
            plan = ev._plan                                                   # built by the evaluation above
            row_of = None
            if plan is not None:
                row_of = torch.full((U,), -1, dtype=torch.int32, device=dev)
                row_of[ub.long()] = torch.arange(nb, dtype=torch.int32, device=dev)
            filt = ev._filter if getattr(ev, "search_used", "fp32") in ("bf16", "int8") else None
            n_keep = min(top_k + 1 + ev.extra_tiles, 63) if filt is not None else top_k + 1
            per = torch.empty((nb, 5 * top_k), dtype=torch.float32, device=dev)
            flg = torch.zeros(nb, dtype=torch.int32, device=dev)

            def level1_fp32():
                return ev._gemm.tile_maxima(eu, ub, trc, plan=plan, row_of=row_of)

            def level1_search():
                return ev._gemm.tile_maxima(eu, ub, trc, plan=plan, row_of=row_of, filt=filt)

            def level2(M, eps=None):
                E.eval_tiles(M, eu, ev._gemm, ub, trc, tec, [1, 2, 4, 3, 5], top_k, per, flg, eps=eps,
                             n_keep=n_keep if eps is not None else None)
            if filt is not None:
                filt.prepare(ei)
                M, eps = level1_search()
            else:
                M, eps = level1_fp32(), None
            level2(M, eps)
            ev_ = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            torch.cuda.synchronize()
            ev_[0].record()
            for _ in range(3):
                ev._gemm.tile_maxima(eu, ub, trc)                            # strikes inside the scoring loop (r01/r02 form)
            ev_[1].record()
            for _ in range(3):
                M32 = level1_fp32()                                          # fp32 MFMA loop + planned fix-up (r03 form)
            ev_[2].record()
            for _ in range(3):
                if filt is not None:
                    M, eps = level1_search()                                 # bf16 bounded filter + the same fix-up (r04)
            ev_[3].record()
            if filt is None:
                M = M32
            for _ in range(3):
                level2(M, eps)
            ev_[4].record()
            torch.cuda.synchronize()
            t_inloop, t_fp32, t_filter, t_rank = (ev_[i].elapsed_time(ev_[i + 1]) / 3e3 for i in range(4))   # seconds
            t_score = t_filter if filt is not None else t_fp32
            flops = 2.0 * I * d_e * nb
            tiles = 2 * ((I + 63) // 64)
            # level 2, algorithmic HBM bytes: per user its tile maxima and factor row read and M*K metrics
            # written; the k-major item copy and the train/test lists read once.
            rank_bytes = nb * (tiles * 4 + d_e * 4 + 5 * top_k * 4) + I * d_e * 4 + \
                (int(train.indptr[-1]) + int(test.indptr[-1])) * 4
            rescore_flops = 2.0 * nb * n_keep * 32 * d_e
            if filt is not None:
                i8 = filt.arith == "int8"
                peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_BF16_PEAK_TFLOPS
                sustained = 3500.0 if i8 else 1300.0
                eval_info["roofline"] = {
                    "bound": "mfma",
                    "kernel": ("split_rows_i8_kernel + tilemax_i8_kernel<%d> (bounded filter: 15-bit fixed point, three int8 "
                               "MFMA products per score in exact int32 accumulators, the bound derived from the "
                               "quantisation) + tilemax_fix_kernel<32> (planned (user, tile) pairs in fp32)"
                               % ((d_e + 31) // 32) if i8 else
                               "split_rows_kernel + tilemax_bf16_kernel<%d> (bounded filter: three bf16 MFMA "
                               "terms per product, fp32 accumulate) + tilemax_fix_kernel<32> (planned "
                               "(user, tile) pairs in fp32)" % ((d_e + 15) // 16)),
                    "users": nb, "ms": t_filter * 1e3, "arith": filt.arith,
                    "achieved": 3.0 * flops / t_filter / 1e12, "peak": peak, "unit": "TOP/s" if i8 else "TFLOP/s",
                    "frac": 3.0 * flops / t_filter / 1e12 / peak,
                    "achieved_note": ("int8 operations ISSUED (3 x 2·I·d per user) over the split + filter + fix-up time; peak = "
                                      "2 x the bf16 dense rate (the guide has no int8 spec row; its micro-benchmark ceiling is "
                                      "3,944); back-to-back v_mfma_i32_32x32x32_i8 on every CU with random operand bits "
                                      "sustain 3,500 (46 clk each: profiles/r04_exp_mfma_valu_overlap.txt), "
                                      "frac_of_sustained prices against that" if i8 else
                                      "bf16 flops ISSUED (3 x 2·I·d per user) over the split + filter + fix-up time; back-to-back "
                                      "v_mfma_f32_32x32x16_bf16 on every CU with random operand bits sustain 1,300 TFLOP/s "
                                      "(60 clk each at the nominal 2.4 GHz instead of 32; 55 with constant operands; "
                                      "profiles/r04_exp_mfma_valu_overlap.txt), frac_of_sustained prices against that"),
                    "frac_of_sustained": 3.0 * flops / t_filter / 1e12 / sustained,
                    "scores_per_s_as_fp32_tflops": flops / t_filter / 1e12,
                    "fp32_mfma_loop_ms": t_fp32 * 1e3, "fp32_mfma_loop_frac": flops / t_fp32 / 1e12 / MFMA_F32_PEAK_TFLOPS,
                    "flops_per_user": 2.0 * I * d_e, "kappa": filt.kappa, "tiles_rescored_per_user": n_keep,
                    "strikes_in_the_loop_ms": t_inloop * 1e3,
                    "strike_plan_pairs": plan.n_pairs if plan is not None else None,
                    "exactness": "the filter only chooses the tiles; every ranked score is the fp32 fmaf chain's, each row is "
                                 "certified against its error bound or redone from fp32 rows (rows_redone_for_ties)"}
            else:
                eval_info["roofline"] = {
                    "bound": "mfma", "kernel": "gather_transpose_kernel + score_tilemax_kernel<32, false> (no strikes in the "
                                               "loop) + tilemax_fix_kernel<32> (planned (user, tile) pairs recomputed with "
                                               "their strikes)" if plan is not None else "score_tilemax_kernel<32, true>",
                    "users": nb,
                    "achieved": flops / t_score / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / t_score / 1e12 / MFMA_F32_PEAK_TFLOPS, "ms": t_score * 1e3,
                    "flops_per_user": 2.0 * I * d_e,
                    "strikes_in_the_loop_ms": t_inloop * 1e3,
                    "strike_plan_pairs": plan.n_pairs if plan is not None else None}
            if plan is not None:
                # the plan is built once per train matrix, outside every evaluation: its one-off cost
                torch.cuda.synchronize()
                tp = time.perf_counter()
                E.TileStrikePlan(trc, I)
                torch.cuda.synchronize()
                eval_info["strike_plan_build_ms"] = (time.perf_counter() - tp) * 1e3
            eval_info["roofline_topk"] = {
                "bound": "hbm", "kernel": "select_rows_kernel (tile maxima -> tile lists) + tile_pairs / chunk kernels + "
                                          "rescore_pairs_kernel (fp32 MFMA chain, 32 users of one tile per wave) + "
                                          "rank_compact_kernel (strikes, ranking, item ids, certificate, metrics) "
                                          "(nrhip_eval_tiles_bounded)",
                "achieved": rank_bytes / t_rank / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": rank_bytes / t_rank / 1e9 / HBM_PEAK_GBS, "ms": t_rank * 1e3,
                "bytes": rank_bytes, "rescore_gflop": rescore_flops / 1e9,
                "note": "pruned design: the [users][I] score matrix is never written; the top-K works on the tile maxima "
                        "(%d floats per user, the phase's HBM stream) and %d rescored 32-item tiles per user; the "
                        "rescoring is bucketed by tile, so an item tile is read once per 32 users (r03: once per user, "
                        "5.6 GB through the L2s, 0.43 ms); the phase is a chain of six short launches" % (tiles, n_keep)}

    line = {
        "metric": "BPR triplets/sec (LightGCN-%s)" % args.shape, "value": triplets_per_s,
        "unit": "triplets/s", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "data_note": "train interactions are a synthetic twin of the named shape (neurec_amd/synth.py: gowalla.train is "
                     "absent from the reference tree); test split: %s" % (data_test if not config4 else "none"),
        "config": {"workload": "LightGCN on synthetic %s-shaped interactions (U=%d, I=%d, E=%d), "
                               "%d layers, dim %d, B=%d per GPU, adj=pre, Adam lr=0.01 reg=1e-3"
                               % (args.shape, U, I, train_nnz, args.layers, args.dim, args.batch),
                   "global_batch": comm.world * args.batch,
                   "parallelism": (
                       "colshard%d (every rank holds %d of the %d embedding columns for all nodes and steps on the global "
                       "batch of %d; one all-gather of the per-triplet partial inner products per step: %d B)"
                       % (comm.world, args.dim // comm.world, args.dim, global_batch, 12 * global_batch * comm.world)
                       if colshard else
                       "dp%d (replicated tables; every rank generates the same global epoch stream from the shared seed "
                       "and steps on the global batch of %d: no exchange in training, tables bit-identical on all ranks)"
                       % (comm.world, global_batch) if replicated else
                       "dp%d (replicated tables; per step one all-gather of 12 B/triplet of ids, every rank steps on "
                       "the global batch)" % comm.world if exchange else
                       "rowshard%d (tables row-sharded; %s hop, all-to-all row lookups, owner-local Adam)"
                       % (comm.world, lg.hop) if rowshard else
                       "dp%d (replicated tables, one all-reduce of dL/dE0 per step)" % comm.world)
                   if comm.active else "single GPU"},
        "final_loss": [float(x) for x in loss2.cpu().numpy()], "timed_region": timed_region,
        "epoch_timed": epoch_timed,
        "epoch_amortised": {"value": epoch_amortised, "unit": "triplets/s",
                            "note": "one epoch = %d steps at the measured step time + the per-epoch sampler and "
                                    "batch-plan launches (%.3f ms, HIP events): E / epoch wall time, SURVEY 8d's "
                                    "definition of the metric" % (steps_per_epoch, epoch_ms)},
        "eval": eval_info, "mf": mf_info, "roofline": roofline,
        "device": E.device_info(),
    }
    if config4 and config4_eval:
        try:
            ev4 = _config4_eval(full, comm, trc, I, args.dim, dev, 65536, batch_rows=args.config4_eval_batch)
        except Exception as e:
            ev4 = {"error": "%s: %s" % (type(e).__name__, e)}
        line["eval"] = ev4
    if colshard:
        # the step's ONE exchange measured on its own (not asserted): all-gather of the per-triplet
        # partial products + the rank-order sums, wall clock between device synchronisations, median of 20
        n3 = 3 * global_batch
        parts = torch.rand(n3, device=dev)
        allp = torch.empty((comm.world, n3), dtype=torch.float32, device=dev)
        given = torch.empty(n3, dtype=torch.float32, device=dev)
        ts = []
        for _ in range(25):
            torch.cuda.synchronize(); comm.barrier()
            t1 = time.perf_counter()
            comm.all_gather_rows(parts, allp)
            E.partials_sum(allp, comm.world, n3, given)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        ex_ms = comm.max_float(float(np.median(ts[5:])) * 1e3)
        line["exchange_measured"] = {"ms_per_step": ex_ms, "bytes_per_rank": 4 * n3, "backend": comm.backend,
                                     "share_of_step": ex_ms / (dt / args.steps * 1e3),
                                     "note": "inside the timed step already; gloo stages through the host (two ranks on "
                                             "one GPU in the tests): only backend nccl is the xGMI figure"}
    if comm.active:
        import torch.distributed as dist
        line["rccl_ranks"] = comm.world if dist.get_backend() == "nccl" else 0
        line["dist_backend"] = dist.get_backend()
        # what of the step is partitioned over the ranks and what every rank repeats
        line["redundant_compute"] = {
            "replicated": "everything: every rank runs the whole step on the global batch (no exchange)",
            "triplets": "everything but the sampler: ids are exchanged, every rank runs the whole step",
            "allreduce": "the propagation's full hops (batch-independent) are repeated on every rank; sampler, BPR "
                         "head and the batch-masked hops are partitioned (each rank its B triplets)",
            "rowshard": False,
            "colshard": False}[args.dp_mode]
        if colshard:
            line["redundant_compute_note"] = ("nothing is computed twice: a rank gathers 1/N of every row's bytes; only the "
                                              "CSR indices (8 B per non-zero) are read by every rank")
    if comm.rank == 0 and comm.world == 1 and default_workload and not args.no_eval:
        # ONE rank's share of a W-rank column-sharded job, measured on this GPU (neurec_amd/colshard.py): rank 0's dim/W
        # columns, the global batch of W*B — what every rank of the job computes per step; the job adds ONE all-gather
        # of 12*W*B bytes per rank to it
        from neurec_amd.colshard import ColumnShardedLightGCN
        shares = {}
        for W in (2, 4, 8):
            gB = W * args.batch
            cs = ColumnShardedLightGCN(comm, A, U, I, E0, args.layers, 0.01, 1e-3, gB, rank=0, world=W)
            sW = BprEpochSampler(trc, I, neg_num=1, batch_size=gB, shuffle=True, seed=2018, plan_users=U)
            bsW = [b for b in sW.batches() if b[0].numel() == gB][:40]
            itW = iter(bsW * 4)

            def one_share():
                b = next(itW)
                cs.step(b[0], b[1], b[2], None, plan=b.plan)
            msW = _hip_timed(one_share, 60, 10)
            shares[str(W)] = {"ms_per_step": msW, "global_batch": gB, "columns_per_rank": args.dim // W,
                              "kernel_width": cs.local.d, "exchange_bytes_per_rank": 12 * gB,
                              "triplets_per_sec_if_exchange_were_free": gB / msW * 1e3}
            del cs, sW, bsW
        line["colshard_one_rank_share"] = shares
    if comm.active and not rowshard and not config4:
        # the SAME global batch stepped by ONE GPU alone (no collectives; every rank does it, rank 0 reports): what
        # the N-GPU figure has to be read against — a full-graph LightGCN step costs the same whatever B is, so a
        # single GPU on the N*B batch already gets most of the "scaling" of a replicated run
        gB = comm.world * args.batch
        lg1 = LightGCNEngine(A, U, I, E0, args.layers, 0.01, 1e-3, gB)
        s1 = BprEpochSampler(trc, I, neg_num=1, batch_size=gB, shuffle=True, seed=2018, plan_users=U)
        bs1 = [b for b in s1.batches() if b[0].numel() == gB][:40]
        it1 = iter(bs1 * 4)

        def one():
            b = next(it1)
            lg1.step(b[0], b[1], b[2], None, plan=b.plan)
        ms1 = _hip_timed(one, 60, 10)
        line["same_global_batch_on_1gpu"] = {"value": gB / ms1 * 1e3, "unit": "triplets/s", "ms_per_step": ms1,
                                             "global_batch": gB}
        del lg1, s1, bs1
    if comm.active and colshard and not config4:
        # STRONG scaling next to the weak figure: the global batch FIXED at B — exactly the step one
        # GPU runs, its columns split over the N ranks.  A full-graph LightGCN step costs the same whatever B is, so the
        # weak figure above grows with N by construction; this one says what N GPUs buy on the SAME work.
        from neurec_amd.colshard import ColumnShardedLightGCN
        lgs = ColumnShardedLightGCN(comm, A, U, I, E0, args.layers, 0.01, 1e-3, args.batch)
        ssm = BprEpochSampler(trc, I, neg_num=1, batch_size=args.batch, shuffle=True, seed=2018, plan_users=U)
        bss = [b for b in ssm.batches() if b[0].numel() == args.batch][:64]
        for b in bss[:12]:
            lgs.step(b[0], b[1], b[2], None, plan=b.plan)
        torch.cuda.synchronize(); comm.barrier()
        t0 = time.perf_counter()
        for b in bss[12:]:
            lgs.step(b[0], b[1], b[2], None, plan=b.plan)
        torch.cuda.synchronize(); comm.barrier()
        dts = comm.max_float(time.perf_counter() - t0) / max(len(bss) - 12, 1)
        line["strong_scaling"] = {"value": args.batch / dts, "unit": "triplets/s", "ms_per_step": dts * 1e3,
                                  "global_batch": args.batch, "steps": len(bss) - 12, "partition": "colshard%d" % comm.world,
                                  "note": "global batch fixed at %d (the 1-GPU step's work split over %d ranks by columns); "
                                          "compare with the 1-GPU value of the same command line" % (args.batch, comm.world)}
        del lgs, ssm, bss
    if comm.active:
        line["north_star_partition"] = ("north_star names ROW-sharded tables with all-to-all lookups: --dp-mode rowshard "
                                        "(sharded.ShardedLightGCN), measured here as the rowshard_config4_law* legs; the "
                                        "headline of an N-rank run is colshard because at the gowalla shape (18 MB of "
                                        "tables) a per-hop exchange of rows costs more than the hop")
    if comm.active and not args.no_config4 and not config4:
        # row-sharded tables on the config-4 LAW at a fixed per-GPU slice (scale = N/8 of BASELINE configs[3]): the
        # partitioned mode's own weak-scaling leg — every rank a slice of the users and of the items, RCCL all-gather
        # per hop, all-to-all lookups (north_star's row-shard path); nothing is repeated across ranks
        ev = None
        torch.cuda.empty_cache()
        for key, hop in (("rowshard_config4_law", None), ("rowshard_config4_law_reduce", "reduce")):
            try:
                leg = leg_config4(comm, dev, args.config4_scale * comm.world / 8.0, hop=hop,
                                  eval_users=65536 if (hop is None and not no_eval_legs) else 0,
                                  eval_batch=args.config4_eval_batch)
            except Exception as e:      # a secondary leg must not take the headline (already measured above) down
                leg = {"error": "%s: %s" % (type(e).__name__, e)}
            if comm.rank == 0:
                line[key] = leg
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
            if mf_info is not None and "eval" in mf_info:
                # the BPR-MF tables of the mf leg through the same comparison
                from oracle import native, ref
                P_h, Q_h = mf.P.cpu().numpy(), mf.Q.cpu().numpy()
                truth = [test.indices[test.indptr[u]:test.indptr[u + 1]].tolist() for u in sample_users]
                S = np.ascontiguousarray(np.matmul(P_h[sample_users], Q_h.T), dtype=np.float32)
                native.mask_train(S, sample_users, train.indptr.astype(np.int64), train.indices)
                fn = ref.eval_matrix if ref.available() else native.eval_matrix
                want = float(np.mean(fn(S, truth, [1, 2, 4, 3, 5], 20, threads=8), axis=0)[2 * 20 + 9])
                got = ev.evaluate_factors(mf.P, mf.Q, torch.from_numpy(sample_users).to(dev), exact_mean=True)
                mf_info["ndcg10_oracle_absdiff"] = abs(float(got[2 * 20 + 9]) - want)
        line["cpu_baseline"] = cb
    else:
        line["cpu_baseline"] = None
    # ---------------- the other BASELINE configs on this GPU (driver-visible legs)
    if comm.rank == 0 and comm.world == 1 and not config4 and default_workload:
        if not args.no_config5:
            line["ngcf"] = leg_ngcf(train, test, trc, tec, dev, not args.no_cpu_baseline)
            line["multivae"] = leg_multivae(train, test, trc, tec, dev, not args.no_cpu_baseline)
    if comm.rank == 0 and comm.world == 1 and not config4 and default_workload and not args.no_mf:
        # BASELINE configs[0] (conf/MF.properties on ml-100k, the reference's own CPU-runnable case): the real split of
        # dataset/ml-100k.rating under the default NeuRec.properties (ratio 0.8) as committed with the golden epoch
        # (tests/golden/tfgraph_ml100k_mf_epoch.npz — the reference tree does not travel), else the synthetic twin
        import scipy.sparse as sp
        fx = os.path.join(ROOT, "tests", "golden", "tfgraph_ml100k_mf_epoch.npz")
        if os.path.isfile(fx):
            z = np.load(fx)
            shp = (int(z["n_users"]), int(z["n_items"]))
            csr = lambda ptr, idx: sp.csr_matrix((np.ones(len(idx), np.float32), idx, ptr), shape=shp)
            tr1, te1 = csr(z["train_indptr"], z["train_indices"]), csr(z["test_indptr"], z["test_indices"])
            src = "the reference's real ml-100k split (80,367 train / 19,633 test pairs; fixture of the golden epoch)"
        else:
            tr1, te1 = synth.interactions("ml-100k", seed=2018)
            src = "synthetic ml-100k-shaped twin (neurec_amd/synth.py)"
        try:
            line["ml100k"], _ = leg_mf(tr1, te1, E.DeviceCSR.from_scipy(tr1), E.DeviceCSR.from_scipy(te1), dev,
                                       args.eval_batch, args.eval_mode, not args.no_eval, not args.no_cpu_baseline,
                                       label="ml-100k")
            line["ml100k"]["data"] = src
        except Exception as e:                                # a secondary leg must not take the headline down
            line["ml100k"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if comm.rank == 0 and comm.world == 1 and not config4 and default_workload and not args.no_config4:
        ev = mf_ev = None
        torch.cuda.empty_cache()
        line["config4"] = leg_config4(comm, dev, args.config4_scale, eval_users=0 if args.no_eval else 65536,
                                      eval_batch=args.config4_eval_batch)
        try:
            line["config4"]["partitions"] = leg_config4_partitions(dev, args.config4_scale)
        except Exception as e:                                # a measurement leg must not take the headline down
            line["config4"]["partitions"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if comm.rank == 0:
        full = os.environ.get("NEUREC_BENCH_FULL") or os.path.join(
            ROOT, "gpurun_out" if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "", "bench_full.json")
        try:
            with open(full, "w") as f:
                json.dump(line, f)
        except OSError:
            pass
        print(json.dumps(line if args.full_line else compact_line(line)))
    comm.shutdown()


if __name__ == "__main__":
    main()
