"""Column-sharded LightGCN (neurec_amd/colshard.py): every rank holds d/W columns of the table for all nodes and
steps on the whole global batch; the one exchange of a step is the all-gather of the per-triplet partial inner
products.  At W = 1 the split step must be the single-GPU step bit for bit; two ranks sharing the one visible GPU
(gloo, host-staged) must reproduce it within fp32 rounding of the inner products and agree with EACH OTHER exactly."""
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


def _setup(d, adj_type="pre"):
    from neurec_amd import graph, synth
    tr, _ = synth.interactions("ml-100k", seed=11)
    coo = tr.tocoo()
    U, I = tr.shape
    A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, adj_type)
    E0 = synth.xavier_uniform(U + I, d, np.random.RandomState(3))
    return A, E0, U, I


def _batches(U, I, B, steps):
    rng = np.random.RandomState(9)
    out = []
    for k in range(steps):
        n = B if k < steps - 1 else B - 37                    # a short last batch
        out.append(tuple(rng.randint(0, x, n).astype(np.int32) for x in (U, I, I)))
    return out


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("d,L,adj", [(64, 3, "pre"), (64, 2, "norm"), (128, 3, "pre"), (16, 3, "pre"), (40, 2, "pre")])
def test_one_rank_is_the_single_gpu_step_bit_for_bit(d, L, adj):
    import torch
    from neurec_amd import parallel
    from neurec_amd.colshard import ColumnShardedLightGCN
    from neurec_amd.trainer import LightGCNEngine
    A, E0, U, I = _setup(d, adj)
    cs = ColumnShardedLightGCN(parallel.Comm(), A, U, I, E0, L, 0.01, 1e-3, 256)
    lg = LightGCNEngine(A, U, I, E0, L, 0.01, 1e-3, 256)
    la, lb = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    for u, p, n in _batches(U, I, 256, 4):
        cs.step(_dev(u), _dev(p), _dev(n), la)
        lg.step(_dev(u), _dev(p), _dev(n), lb)
        np.testing.assert_array_equal(la.cpu().numpy(), lb.cpu().numpy())
    np.testing.assert_array_equal(cs.local.E0.cpu().numpy(), lg.E0.cpu().numpy())
    np.testing.assert_array_equal(cs.local.m.cpu().numpy(), lg.m.cpu().numpy())
    eu, ei = cs.final_embeddings()
    fu, fi = lg.final_embeddings()
    np.testing.assert_array_equal(eu.cpu().numpy(), fu.cpu().numpy())


def _worker(rank, world, port, out, d, L):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), NEUREC_DIST_BACKEND="gloo")
    from neurec_amd import parallel
    from neurec_amd.colshard import ColumnShardedLightGCN
    comm = parallel.init_from_env()
    A, E0, U, I = _setup(d)
    cs = ColumnShardedLightGCN(comm, A, U, I, E0, L, 0.01, 1e-3, 256)
    loss = torch.zeros(2, device="cuda")
    losses = []
    for u, p, n in _batches(U, I, 256, 4):
        cs.step(_dev(u), _dev(p), _dev(n), loss)
        losses.append(loss.cpu().numpy().copy())
    eu, ei = cs.final_embeddings()
    np.savez(os.path.join(out, "rank%d.npz" % rank), losses=np.asarray(losses), table=cs.table().cpu().numpy(),
             eu=eu.cpu().numpy(), ei=ei.cpu().numpy(), exchange=cs.exchange_bytes_per_step)
    comm.shutdown()


@pytest.mark.parametrize("d,L", [(64, 3), (32, 2)])
def test_two_ranks_equal_one_gpu_within_inner_product_rounding(tmp_path, d, L):
    import torch
    import torch.multiprocessing as mp
    from neurec_amd.trainer import LightGCNEngine
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), d, L), nprocs=world, join=True, start_method="spawn")
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    # the ranks agree with each other exactly: same summed inner products, same loss, same gathered E*
    np.testing.assert_array_equal(r[0]["losses"], r[1]["losses"])
    np.testing.assert_array_equal(r[0]["eu"], r[1]["eu"])
    np.testing.assert_array_equal(r[0]["ei"], r[1]["ei"])
    assert int(r[0]["exchange"]) == 2 * 3 * (256 - 37) * 4          # 12 B per triplet and rank: the last batch's
    # ... and with the single-GPU engine on the same batches within the rounding of x_b = sum of partial dots
    A, E0, U, I = _setup(d)
    lg = LightGCNEngine(A, U, I, E0, L, 0.01, 1e-3, 256)
    loss = torch.zeros(2, device="cuda")
    want = []
    for u, p, n in _batches(U, I, 256, 4):
        lg.step(_dev(u), _dev(p), _dev(n), loss)
        want.append(loss.cpu().numpy().copy())
    want = np.asarray(want)
    assert np.abs(r[0]["losses"] - want).max() <= 1e-6 * np.abs(want).max()
    full = np.concatenate([r[0]["table"], r[1]["table"]], axis=1)
    got = lg.E0.cpu().numpy()
    # Adam turns a last-ulp change of a gradient near zero into a fraction of a step; 4 steps at lr 0.01
    assert np.abs(full - got).max() <= 2e-4 and np.mean(np.abs(full - got) > 1e-6) < 0.01
    fu, fi = lg.final_embeddings()
    assert np.abs(r[0]["eu"] - fu.cpu().numpy()).max() <= 2e-4


def test_one_rank_share_of_8_costs_no_more_than_1p2x_the_share_of_4():
    """the default `--gpus 8` path steps on the global batch of 8 x 1,024 with 8 of the 64 columns; its
    one-rank step (no exchange) must stay within 1.2x of the 4-rank share's (16 columns, 4 x 1,024) at the gowalla
    shape — HIP events, best of three, NOT under a profiler (kernel tracing made this host-bound loop read 0.41 ms
    where the untraced run reads 0.157: profiles/r03_bench.json vs BENCH_r03.json)."""
    import torch
    from neurec_amd import engine as E, graph, parallel, synth
    from neurec_amd.colshard import ColumnShardedLightGCN
    from neurec_amd.trainer import BprEpochSampler
    tr, _ = synth.interactions("gowalla", seed=2018)
    U, I = tr.shape
    coo = tr.tocoo()
    A = graph.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
    E0 = synth.xavier_uniform(U + I, 64, np.random.RandomState(2017))
    trc = E.DeviceCSR.from_scipy(tr)
    ms = {}
    for W in (4, 8):
        gB = W * 1024
        cs = ColumnShardedLightGCN(parallel.Comm(), A, U, I, E0, 3, 0.01, 1e-3, gB, rank=0, world=W)
        s = BprEpochSampler(trc, I, batch_size=gB, seed=2018, plan_users=U)
        bs = [b for b in s.batches() if b[0].numel() == gB][:30]
        best = 1e9
        for _ in range(3):
            for b in bs[:10]:
                cs.step(b[0], b[1], b[2], None, plan=b.plan)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for b in bs:
                cs.step(b[0], b[1], b[2], None, plan=b.plan)
            e.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(e) / len(bs))
        ms[W] = best
        del cs, s, bs
    print("one-rank share: W=4 %.4f ms, W=8 %.4f ms" % (ms[4], ms[8]))
    assert ms[8] <= 1.2 * ms[4], ms
