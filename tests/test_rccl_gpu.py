"""RCCL (torch.distributed backend "nccl") on the ONE GPU of the test box: a process group of world size 1 whose
collectives are FORCED through the library (parallel.Comm(force=True) / NEUREC_DIST_FORCE_GROUP=1).  What it proves:
RCCL loads and initialises next to the ctypes-launched HIP kernels, every `Comm` method's call signature, dtype and
split arguments are accepted by ProcessGroupNCCL, and every sharded engine's step — with its collectives really issued
(counted) on RCCL's stream and its kernels on torch's current stream — gives the SAME BITS as the same engine without a
process group, with and without NEUREC_DIST_DEBUG_SYNC (a device synchronisation around every collective): a missing
stream dependency between a collective and the kernel that reads its output would show as a difference.  What it
cannot prove: anything about two ranks (the two-rank runs are gloo: tests/test_sharded_gpu.py, test_parallel_gpu.py).
Reference behaviour the engines keep: LightGCN.py:132-149 (a hop), MF.py:54-76, NGCF.py:160-202."""
import json
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _graph(d):
    from neurec_amd import graph, synth
    tr, _ = synth.interactions("ml-100k", seed=11)
    coo = tr.tocoo()
    U, I = tr.shape
    E0 = synth.xavier_uniform(U + I, d, np.random.RandomState(3))
    return tr, coo, E0, U, I


def _batches(U, I, B, steps, seed=9):
    rng = np.random.RandomState(seed)
    return [(rng.randint(0, U, B).astype(np.int32), rng.randint(0, I, B).astype(np.int32),
             rng.randint(0, I, B).astype(np.int32)) for _ in range(steps)]


def _comm_methods(comm, torch):
    """every Comm method once, with the dtypes and shapes the engines use"""
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x = torch.rand(37, 48, generator=g, device=dev)
    out = torch.empty_like(x)
    assert comm.all_gather_rows(x, out) is out and torch.equal(out, x)
    flag = (torch.rand(101, generator=g, device=dev) < 0.5).to(torch.uint8)
    fo = torch.empty_like(flag)
    comm.all_gather_rows(flag, fo)
    assert torch.equal(fo, flag)
    out.zero_()
    tok = comm.all_gather_rows_start(x, out)                       # the column-sliced hop's asynchronous form
    y = x * 2.0                                                    # work on the current stream under the collective
    comm.all_gather_rows_finish(tok)
    assert torch.equal(out, x) and torch.equal(y, x * 2.0)
    buf = x.clone()
    comm.bcast_rows_finish(comm.bcast_rows_start(buf, 0))
    assert torch.equal(buf, x)
    # variable all-to-all: rows of float32 [n][2d] and of int32 [n][2] (the routed (row, code) pairs), the counts
    # exchanged (recv_counts None) and given; an empty send
    for send in (x, torch.arange(74, dtype=torch.int32, device=dev).view(37, 2), x[:0]):
        n = send.shape[0]
        got, rc = comm.all_to_all_rows(send, [n])
        assert rc == [n] and torch.equal(got, send)
        got, rc = comm.all_to_all_rows(send, [n], [n])
        assert rc == [n] and torch.equal(got, send)
    ids = [torch.arange(k, k + 16, dtype=torch.int32, device=dev) for k in (0, 100, 200)]
    parts = comm.allgather_cat_finish(comm.allgather_cat_start(ids))
    assert all(torch.equal(a, b) for a, b in zip(parts, ids))
    for dt in (torch.float32, torch.float64):
        t = torch.arange(9, dtype=dt, device=dev)
        assert comm.allreduce_sum_(t) is t and torch.equal(t, torch.arange(9, dtype=dt, device=dev))
    assert torch.equal(comm.broadcast_(x.clone()), x)
    assert comm.max_float(2.5) == 2.5
    comm.barrier()


def _worker(rank, port, out_path):
    import torch
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      NEUREC_DIST_BACKEND="nccl", NEUREC_DIST_FORCE_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from neurec_amd import graph, parallel
    from neurec_amd.colshard import ColumnShardedLightGCN
    from neurec_amd.sharded import ShardedLightGCN, ShardedMF
    from neurec_amd.sharded_ngcf import ShardedNGCF
    comm = parallel.init_from_env()
    assert comm.live and not comm.active and comm.backend == "nccl" and dist.get_backend() == "nccl"
    plain = parallel.Comm()                                         # no process group: the collectives are copies
    synced = parallel.Comm(0, 1, 0, "nccl", force=True)
    synced.debug_sync = True                                        # a device synchronisation around every collective
    report = {"backend": dist.get_backend(), "nccl_version": list(torch.cuda.nccl.version())}
    _comm_methods(comm, torch)
    _comm_methods(synced, torch)
    report["comm_calls"] = dict(comm.calls)
    # r06: the blocking all-gather goes through librccl's C API on the CURRENT stream (parallel._RcclDirect); the
    # process group's own collective is the fallback and gives the same bits
    report["direct_all_gather"] = comm._direct is not None
    os.environ["NEUREC_RCCL_DIRECT"] = "0"
    via_torch = parallel.Comm(0, 1, 0, "nccl", force=True)
    x = torch.rand(257, 12, device="cuda")
    a, b = torch.empty_like(x), torch.empty_like(x)
    comm.all_gather_rows(x, a)
    via_torch.all_gather_rows(x, b)
    report["direct_equals_process_group"] = bool(torch.equal(a, b) and torch.equal(a, x)) and via_torch._direct is None
    del os.environ["NEUREC_RCCL_DIRECT"]

    cuda = lambda b: tuple(torch.from_numpy(x).cuda() for x in b)

    def lightgcn(c, adj_type, d, **kw):
        tr, coo, E0, U, I = _graph(d)
        A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, adj_type)
        eng = ShardedLightGCN(c, A, U, I, E0, 3, 0.01, 1e-3, 128, **kw)
        steps = _batches(U, I, 128, 3)
        eng.plan_epoch(*(torch.from_numpy(np.concatenate([s[k] for s in steps])).cuda() for k in range(3)), 128)
        losses = []
        for k, b in enumerate(steps):
            l2 = torch.zeros(2, device="cuda")
            eng.step(*cuda(b), l2, batch_index=k)
            losses.append(l2.cpu().numpy())
        eu, ei = eng.final_embeddings()
        return [eng.table_rows().cpu().numpy(), np.asarray(losses), eu.cpu().numpy(), ei.cpu().numpy()]

    def mf(c):
        tr, coo, E0, U, I = _graph(64)
        eng = ShardedMF(c, E0[:U], E0[U:], 0.001, 0.01, 128)
        for b in _batches(U, I, 128, 3):
            eng.step(*cuda(b), torch.zeros(2, device="cuda"))
        return [t.cpu().numpy() for t in eng.tables()]

    def ngcf(c):
        from neurec_amd.graph import ngcf_adjacency, transpose_csr
        tr, coo, _, U, I = _graph(16)
        A = ngcf_adjacency(tr, "norm")
        rng = np.random.RandomState(5)
        widths = (16, 16, 16)
        E0 = (rng.randn(U + I, 16) * 0.1).astype(np.float32)
        W = [((rng.randn(wi, wo) * 0.3).astype(np.float32), (rng.randn(1, wo) * 0.05).astype(np.float32),
              (rng.randn(wi, wo) * 0.3).astype(np.float32), (rng.randn(1, wo) * 0.05).astype(np.float32))
             for wi, wo in zip(widths[:-1], widths[1:])]
        eng = ShardedNGCF(c, A, transpose_csr(A), U, I, E0, W, 0.005, 1e-4, 0.1, 96)
        masks = [[(rng.rand(U + I, w) < 0.9).astype(np.uint8) for w in widths[1:]] for _ in range(3)]
        for s, b in enumerate(_batches(U, I, 96, 3, seed=17)):
            eng.step(*cuda(b), torch.zeros(2, device="cuda"), masks=eng.local_masks(masks[s]))
        return [eng.ego_table().cpu().numpy()] + [eng.W[k][j].cpu().numpy() for k in range(2) for j in range(4)]

    def colshard(c):
        from neurec_amd import engine as E
        from neurec_amd.trainer import BprEpochSampler
        tr, coo, E0, U, I = _graph(64)
        A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
        eng = ColumnShardedLightGCN(c, A, U, I, E0, 3, 0.01, 1e-3, 128)
        sam = BprEpochSampler(E.DeviceCSR.from_scipy(tr), I, neg_num=1, batch_size=128, shuffle=True, seed=7,
                              plan_users=U)
        for k, b in enumerate(sam.batches()):
            if k == 3:
                break
            eng.step(b[0], b[1], b[2], None, plan=b.plan)
        eu, ei = eng.final_embeddings()
        return [eng.table().cpu().numpy(), eu.cpu().numpy(), ei.cpu().numpy()]

    cases = {"lightgcn_pre64_allgather": lambda c: lightgcn(c, "pre", 64, hop="allgather"),
             "lightgcn_norm64_sliced": lambda c: lightgcn(c, "norm", 64, hop="sliced", col_slices=2),
             "lightgcn_pre128_sliced": lambda c: lightgcn(c, "pre", 128, hop="sliced", col_slices=2),
             "lightgcn_pre128_reduce": lambda c: lightgcn(c, "pre", 128, hop="reduce"),
             "mf": mf, "ngcf": ngcf, "colshard": colshard}
    report["engines"] = {}
    for name, run in cases.items():
        before = dict(comm.calls)
        got = run(comm)
        issued = {k: v - before.get(k, 0) for k, v in comm.calls.items() if v != before.get(k, 0)}
        want = run(plain)
        slow = run(synced)
        report["engines"][name] = {
            "collectives_issued": issued,
            "equals_no_group": bool(all(np.array_equal(a, b) for a, b in zip(got, want))),
            "equals_debug_sync": bool(all(np.array_equal(a, b) for a, b in zip(got, slow))),
            "finite": bool(all(np.isfinite(a).all() for a in got))}
    with open(out_path, "w") as f:
        json.dump(report, f)
    comm.barrier()
    comm.shutdown()


def test_every_collective_and_every_sharded_engine_over_rccl_at_world_size_one(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl.json")
    mp.start_processes(_worker, args=(_free_port(), out), nprocs=1, join=True, start_method="spawn")
    with open(out) as f:
        rep = json.load(f)
    print("RCCL at world size 1:", json.dumps(rep))
    assert rep["backend"] == "nccl"
    assert rep["direct_all_gather"] and rep["direct_equals_process_group"]
    for name in ("all_gather", "all_gather_async", "broadcast", "all_to_all", "all_gather_ids", "all_reduce"):
        assert rep["comm_calls"].get(name, 0) > 0, name
    for name, r in rep["engines"].items():
        assert r["collectives_issued"], name                      # the step really went through RCCL
        assert r["finite"], name
        # the reduce form sums per-rank partials: at one rank that is one partial, still the same bits
        assert r["equals_no_group"], name
        assert r["equals_debug_sync"], name
