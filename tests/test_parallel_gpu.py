"""The N>1 path with real HIP kernels: two ranks (sharing the one visible GPU, exchanging over
gloo because RCCL refuses two ranks on one device) run the data-parallel LightGCN step and the
sharded evaluation; the result must equal one process with batch = 2·B.  On a multi-GPU node the
only difference is the transport (backend "nccl" = RCCL over xGMI)."""
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


def _setup():
    from neurec_amd import graph, synth
    tr, te = synth.interactions("ml-100k", seed=7)
    coo = tr.tocoo()
    U, I = tr.shape
    A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
    E0 = synth.xavier_uniform(U + I, 64, np.random.RandomState(1))
    return tr, te, A, E0, U, I


def _worker(rank, world, port, out, mode="allreduce"):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NEUREC_DIST_BACKEND="gloo")
    from neurec_amd import engine as E, parallel
    from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, LightGCNEngine
    comm = parallel.init_from_env()
    tr, te, A, E0, U, I = _setup()
    lg = LightGCNEngine(A, U, I, E0, 2, 0.01, 1e-3, 256 * (world if mode == "triplets" else 1))
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    sampler = BprEpochSampler(trc, I, batch_size=256, seed=5, rank=rank, world=world)
    it = sampler.batches()
    if mode == "triplets":                   # ids all-gathered one step ahead, global-batch step
        token = comm.allgather_cat_start(next(it))
        for s in range(4):
            cur = token
            if s < 3:
                token = comm.allgather_cat_start(next(it))
            bu, bp, bn = comm.allgather_cat_finish(cur)
            assert bu.numel() == 256 * world
            lg.step(bu, bp, bn, None)
    else:
        for _ in range(4):
            bu, bp, bn = next(it)
            lg.step(bu, bp, bn, None, grad_sync=comm.allreduce_sum_)
    users = np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)
    mine = torch.from_numpy(parallel.shard_users(users, rank, world)).cuda()
    eu, ei = lg.final_embeddings()
    ev = FullRankEvaluator(trc, tec, [2, 4], 10, batch_rows=512)
    sums = torch.from_numpy(ev.evaluate_factors(eu.contiguous(), ei.contiguous(), mine) * mine.numel()).cuda()
    comm.allreduce_sum_(sums)
    if rank == 0:
        np.savez(out, E0=lg.E0.cpu().numpy(), means=(sums / len(users)).cpu().numpy())
    comm.barrier()
    comm.shutdown()


@pytest.mark.parametrize("mode", ["allreduce", "triplets"])
def test_two_ranks_equal_one_process_with_doubled_batch(tmp_path, mode):
    import torch
    import torch.multiprocessing as mp
    from neurec_amd import engine as E
    from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, LightGCNEngine
    out = str(tmp_path / "r0.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True,
                       start_method="spawn")
    got = np.load(out)
    tr, te, A, E0, U, I = _setup()
    lg = LightGCNEngine(A, U, I, E0, 2, 0.01, 1e-3, 512)
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    halves = [BprEpochSampler(trc, I, batch_size=256, seed=5, rank=r, world=2).batches() for r in range(2)]
    for _ in range(4):
        parts = [next(h) for h in halves]                       # the two ranks' batches, concatenated
        bu, bp, bn = (torch.cat([p[k] for p in parts]).contiguous() for k in range(3))
        lg.step(bu, bp, bn, None)
    assert np.abs(got["E0"] - lg.E0.cpu().numpy()).max() < 1e-5
    users = torch.from_numpy(np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)).cuda()
    eu, ei = lg.final_embeddings()
    want = FullRankEvaluator(trc, tec, [2, 4], 10, batch_rows=512).evaluate_factors(
        eu.contiguous(), ei.contiguous(), users)
    np.testing.assert_allclose(got["means"], want, atol=2e-3)   # tables differ by ~1e-6 -> rare rank swaps
