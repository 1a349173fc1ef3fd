"""Evaluation of a row-sharded model (SURVEY 8e "Evaluator"): every rank ranks ITS users against the
all-gathered ITEM table (ShardedLightGCN.eval_factors — the user table never moves) and the metric sums are added
over the ranks once.  Two ranks sharing the one visible GPU (gloo, host-staged) must reproduce the single-process
engine + FullRankEvaluator: the factor rows and every user's metric row bit for bit, the means to the last bits of an
fp64 sum taken in another order."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MIDS, K = [1, 2, 4, 3, 5], 20


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(d):
    from neurec_amd import graph, synth
    tr, te = synth.interactions("ml-100k", seed=11)
    coo = tr.tocoo()
    U, I = tr.shape
    A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
    E0 = synth.xavier_uniform(U + I, d, np.random.RandomState(3))
    return tr, te, A, E0, U, I


def _batches(U, I, world, B, steps):
    rng = np.random.RandomState(9)
    return [[(rng.randint(0, U, B).astype(np.int32), rng.randint(0, I, B).astype(np.int32),
              rng.randint(0, I, B).astype(np.int32)) for _ in range(world)] for _ in range(steps)]


def _worker(rank, world, port, out, d, hop):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NEUREC_DIST_BACKEND="gloo")
    from neurec_amd import engine as E, parallel
    from neurec_amd.sharded import ShardedEvaluator, ShardedLightGCN
    comm = parallel.init_from_env()
    tr, te, A, E0, U, I = _setup(d)
    eng = ShardedLightGCN(comm, A, U, I, E0, 3, 0.01, 1e-3, 128, hop=hop)
    for step in _batches(U, I, world, 128, 2):
        bu, bp, bn = (torch.from_numpy(x).cuda() for x in step[rank])
        eng.step(bu, bp, bn, None)
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    ev = ShardedEvaluator(comm, trc.rows(eng.ulo, eng.uhi), tec.rows(eng.ulo, eng.uhi), MIDS, K, batch_rows=256)
    calls = dict(comm.calls)
    means = ev.evaluate(eng)
    # what an evaluation exchanged: the hops' gathers, ONE all-gather of the item blocks, ONE all-reduce of the sums
    gathers = comm.calls.get("all_gather", 0) - calls.get("all_gather", 0)
    reduces = comm.calls.get("all_reduce", 0) - calls.get("all_reduce", 0)
    eu, items = eng.eval_factors()
    rows = ev.ev.evaluate_factors(eu, items, ev.users, per_user=True)
    np.savez(out % rank, means=means, eu=eu.cpu().numpy(), items=items.cpu().numpy(), rows=np.asarray(rows),
             users=ev.users.cpu().numpy() + eng.ulo, n_total=ev.n_total, gathers=gathers, reduces=reduces)
    comm.barrier()
    comm.shutdown()


@pytest.mark.parametrize("d,hop", [(64, "sliced"), (128, "sliced"), (64, "allgather")])
def test_two_rank_evaluation_equals_the_single_engine(tmp_path, d, hop):
    import torch
    import torch.multiprocessing as mp
    from neurec_amd import engine as E, parallel
    from neurec_amd.trainer import FullRankEvaluator, LightGCNEngine
    out = str(tmp_path / "r%d.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out, d, hop), nprocs=2, join=True, start_method="spawn")
    got = [np.load(out % r) for r in range(2)]
    tr, te, A, E0, U, I = _setup(d)
    lg = LightGCNEngine(A, U, I, E0, 3, 0.01, 1e-3, 256)
    for step in _batches(U, I, 2, 128, 2):
        bu, bp, bn = (torch.from_numpy(np.concatenate([s[k] for s in step])).cuda() for k in range(3))
        lg.step(bu, bp, bn, None)
    eu, ei = lg.final_embeddings()
    eu, ei = eu.contiguous(), ei.contiguous()
    part = parallel.BipartitePartition(U, I, 2)
    for r in range(2):
        lo, hi = part.users_of(r)
        np.testing.assert_array_equal(got[r]["eu"], eu[lo:hi].cpu().numpy())      # a rank holds ITS user rows only
        np.testing.assert_array_equal(got[r]["items"], ei.cpu().numpy())          # ... and the whole item table
    trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
    users = np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)
    ev = FullRankEvaluator(trc, tec, MIDS, K, batch_rows=256)
    ud = torch.from_numpy(users).cuda()
    rows = np.asarray(ev.evaluate_factors(eu, ei, ud, per_user=True))
    np.testing.assert_array_equal(np.concatenate([got[0]["users"], got[1]["users"]]), users)
    np.testing.assert_array_equal(np.concatenate([got[0]["rows"], got[1]["rows"]]), rows)
    means = ev.evaluate_factors(eu, ei, ud)
    assert int(got[0]["n_total"]) == len(users)
    np.testing.assert_array_equal(got[0]["means"], got[1]["means"])
    np.testing.assert_allclose(got[0]["means"], means, rtol=0, atol=1e-13)
    assert means[2 * K + 9] > 0                                                    # NDCG@10 of a model, not zeros
    for r in range(2):
        assert int(got[r]["reduces"]) == 1 and int(got[r]["gathers"]) == 1      # the sums; the item blocks
