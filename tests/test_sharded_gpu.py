"""Row-sharded LightGCN (BASELINE config 4 path) with real HIP kernels: two ranks sharing the one
visible GPU (collectives over gloo, host-staged — RCCL refuses two ranks on one device) must
reproduce the single-process engine stepping on the concatenated global batch.  On a multi-GPU
node only the transport differs (all_gather_into_tensor / all_to_all_single on RCCL)."""
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


def _setup(adj_type, d):
    from neurec_amd import graph, synth
    tr, te = synth.interactions("ml-100k", seed=11)
    coo = tr.tocoo()
    U, I = tr.shape
    A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, adj_type)
    E0 = synth.xavier_uniform(U + I, d, np.random.RandomState(3))
    return tr, A, E0, U, I


def _batches(U, I, world, B, steps):
    rng = np.random.RandomState(9)
    return [[(rng.randint(0, U, B).astype(np.int32), rng.randint(0, I, B).astype(np.int32),
              rng.randint(0, I, B).astype(np.int32)) for _ in range(world)] for _ in range(steps)]


def _worker(rank, world, port, out, adj_type, d):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NEUREC_DIST_BACKEND="gloo")
    from neurec_amd import parallel
    from neurec_amd.sharded import ShardedLightGCN
    comm = parallel.init_from_env()
    tr, A, E0, U, I = _setup(adj_type, d)
    eng = ShardedLightGCN(comm, A, U, I, E0, 2, 0.01, 1e-3, 128)
    losses = []
    for step in _batches(U, I, world, 128, 3):
        bu, bp, bn = (torch.from_numpy(x).cuda() for x in step[rank])
        l2 = torch.zeros(2, device="cuda")
        eng.step(bu, bp, bn, l2)
        comm.allreduce_sum_(l2)
        losses.append(l2.cpu().numpy())
    eu, ei = eng.final_embeddings()
    table = torch.zeros(eng.Npad, d, device="cuda")
    comm.all_gather_rows(eng.E0, table)
    if rank == 0:
        np.savez(out, E0=table[:U + I].cpu().numpy(), losses=np.asarray(losses),
                 eu=eu.cpu().numpy(), ei=ei.cpu().numpy())
    comm.barrier()
    comm.shutdown()


@pytest.mark.parametrize("adj_type,d", [("pre", 64), ("norm", 64), ("pre", 128)])
def test_sharded_lightgcn_equals_single_process(tmp_path, adj_type, d):
    import torch
    import torch.multiprocessing as mp
    from neurec_amd.trainer import LightGCNEngine
    out = str(tmp_path / "r0.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out, adj_type, d), nprocs=2, join=True,
                       start_method="spawn")
    got = np.load(out)
    tr, A, E0, U, I = _setup(adj_type, d)
    lg = LightGCNEngine(A, U, I, E0, 2, 0.01, 1e-3, 256)
    want_losses = []
    for step in _batches(U, I, 2, 128, 3):
        bu, bp, bn = (torch.from_numpy(np.concatenate([s[k] for s in step])).cuda() for k in range(3))
        l2 = torch.zeros(2, device="cuda")
        lg.step(bu, bp, bn, l2)
        want_losses.append(l2.cpu().numpy())
    assert np.abs(got["E0"] - lg.E0.cpu().numpy()).max() < 1e-5
    np.testing.assert_allclose(got["losses"], np.asarray(want_losses), rtol=1e-5)
    eu, ei = lg.final_embeddings()
    assert np.abs(got["eu"] - eu.cpu().numpy()).max() < 1e-5
    assert np.abs(got["ei"] - ei.cpu().numpy()).max() < 1e-5


def _mf_worker(rank, world, port, out):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NEUREC_DIST_BACKEND="gloo")
    from neurec_amd import parallel
    from neurec_amd.sharded import ShardedMF
    comm = parallel.init_from_env()
    rng = np.random.RandomState(21)
    U, I, d = 500, 333, 64
    P0 = (rng.randn(U, d) * 0.05).astype(np.float32)
    Q0 = (rng.randn(I, d) * 0.05).astype(np.float32)
    eng = ShardedMF(comm, P0, Q0, 0.001, 0.01, 96)
    losses = []
    for step in _batches(U, I, world, 96, 4):
        bu, bp, bn = (torch.from_numpy(x).cuda() for x in step[rank])
        l2 = torch.zeros(2, device="cuda")
        eng.step(bu, bp, bn, l2)
        comm.allreduce_sum_(l2)
        losses.append(l2.cpu().numpy())
    P, Q = eng.tables()
    if rank == 0:
        np.savez(out, P=P.cpu().numpy(), Q=Q.cpu().numpy(), losses=np.asarray(losses))
    comm.barrier()
    comm.shutdown()


def test_sharded_mf_equals_single_process(tmp_path):
    """row-sharded BPR-MF (three all-to-alls per step) == MFEngine on the concatenated batch"""
    import torch
    import torch.multiprocessing as mp
    from neurec_amd.trainer import MFEngine
    out = str(tmp_path / "mf.npz")
    mp.start_processes(_mf_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    rng = np.random.RandomState(21)
    U, I, d = 500, 333, 64
    P0 = (rng.randn(U, d) * 0.05).astype(np.float32)
    Q0 = (rng.randn(I, d) * 0.05).astype(np.float32)
    mf = MFEngine(P0, Q0, 0.001, 0.01, 192)
    want_losses = []
    for step in _batches(U, I, 2, 96, 4):
        bu, bp, bn = (torch.from_numpy(np.concatenate([s[k] for s in step])).cuda() for k in range(3))
        l2 = torch.zeros(2, device="cuda")
        mf.step(bu, bp, bn, l2)
        want_losses.append(l2.cpu().numpy())
    np.testing.assert_allclose(got["losses"], np.asarray(want_losses), rtol=1e-5)
    assert np.abs(got["P"] - mf.P.cpu().numpy()).max() < 5e-6
    assert np.abs(got["Q"] - mf.Q.cpu().numpy()).max() < 5e-6
