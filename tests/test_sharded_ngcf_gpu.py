"""Row-sharded NGCF (neurec_amd/sharded_ngcf.py; SURVEY §8e last row) with real HIP kernels: two ranks sharing the one
visible GPU (gloo, host-staged) against the single-GPU engine (ngcf_wide.NGCFWideEngine) stepping on the concatenated
global batch with the same dropout masks.  Every node row sees the same arithmetic; the weight gradients are per-rank
contractions + an all-reduce, so they agree to fp32 rounding (1e-5, north_star), not bit for bit."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(widths):
    from neurec_amd import synth
    from neurec_amd.graph import ngcf_adjacency, transpose_csr
    tr, _ = synth.interactions("ml-100k", seed=11)
    U, I = tr.shape
    A = ngcf_adjacency(tr, "norm")
    rng = np.random.RandomState(5)
    d = widths[0]
    E0 = (rng.randn(U + I, d) * 0.1).astype(np.float32)
    W = []
    for k in range(len(widths) - 1):
        wi, wo = widths[k], widths[k + 1]
        W.append(((rng.randn(wi, wo) * 0.3).astype(np.float32), (rng.randn(1, wo) * 0.05).astype(np.float32),
                  (rng.randn(wi, wo) * 0.3).astype(np.float32), (rng.randn(1, wo) * 0.05).astype(np.float32)))
    return A, transpose_csr(A), E0, W, U, I


def _inputs(U, I, widths, world, B, steps):
    rng = np.random.RandomState(17)
    batches = [[(rng.randint(0, U, B).astype(np.int32), rng.randint(0, I, B).astype(np.int32),
                 rng.randint(0, I, B).astype(np.int32)) for _ in range(world)] for _ in range(steps)]
    masks = [[(rng.rand(U + I, w) < 0.9).astype(np.uint8) for w in widths[1:]] for _ in range(steps + 1)]
    return batches, masks


def _worker(rank, world, port, out, widths):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NEUREC_DIST_BACKEND="gloo")
    from neurec_amd import parallel
    from neurec_amd.sharded_ngcf import ShardedNGCF
    comm = parallel.init_from_env()
    A, At, E0, W, U, I = _setup(widths)
    eng = ShardedNGCF(comm, A, At, U, I, E0, W, 0.005, 1e-4, 0.1, 96)
    batches, masks = _inputs(U, I, widths, world, 96, 3)
    losses = []
    for s, step in enumerate(batches):
        bu, bp, bn = (torch.from_numpy(x).cuda() for x in step[rank])
        l2 = torch.zeros(2, device="cuda")
        eng.step(bu, bp, bn, l2, masks=eng.local_masks(masks[s]))
        comm.allreduce_sum_(l2)
        losses.append(l2.cpu().numpy())
    eu, ei = eng.final_embeddings(eng.local_masks(masks[-1]))
    table = eng.ego_table()
    if rank == 0:
        np.savez(out, E0=table.cpu().numpy(), losses=np.asarray(losses), eu=eu.cpu().numpy(), ei=ei.cpu().numpy(),
                 **{"W%d_%d" % (k, j): eng.W[k][j].cpu().numpy() for k in range(len(W)) for j in range(4)})
    comm.barrier()
    comm.shutdown()


@pytest.mark.parametrize("widths", [(16, 16, 16), (24, 32, 8), (64, 64)])
def test_sharded_ngcf_equals_the_single_engine(tmp_path, widths):
    import torch
    import torch.multiprocessing as mp
    from neurec_amd.ngcf_wide import NGCFWideEngine
    out = str(tmp_path / "r0.npz")
    mp.start_processes(_worker, args=(2, _free_port(), out, widths), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    A, At, E0, W, U, I = _setup(widths)
    eng = NGCFWideEngine(A, At, U, I, E0, W, 0.005, 1e-4, 0.1, 192)
    batches, masks = _inputs(U, I, widths, 2, 96, 3)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    want_losses = []
    for s, step in enumerate(batches):
        bu, bp, bn = (dev(np.concatenate([r[k] for r in step])) for k in range(3))
        l2 = torch.zeros(2, device="cuda")
        eng.step(bu, bp, bn, l2, masks=[dev(m) for m in masks[s]])
        want_losses.append(l2.cpu().numpy())
    out_full = eng.forward([dev(m) for m in masks[-1]]).cpu().numpy()
    err = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
    np.testing.assert_allclose(got["losses"], np.asarray(want_losses), rtol=1e-5)
    dE = err(got["E0"], eng.E0.cpu().numpy())
    dW = max(err(got["W%d_%d" % (k, j)], eng.W[k][j].cpu().numpy().reshape(got["W%d_%d" % (k, j)].shape))
             for k in range(len(W)) for j in range(4))
    dO = max(err(got["eu"], out_full[:U]), err(got["ei"], out_full[U:]))
    print("sharded NGCF %s vs one GPU after 3 steps: ego table %.1e, layer weights %.1e, evaluation tables %.1e"
          % (widths, dE, dW, dO))
    assert dE <= TOL and dW <= TOL and dO <= TOL
