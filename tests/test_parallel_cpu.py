"""Multi-process path on CPU (gloo, world_size 2): the sharding plan and the one exchange step
of neurec_amd/parallel.py.  The HIP kernels cannot run here, so the per-rank compute is played
by the CPU oracle — inside this test only — which is enough to prove the distributed algebra:
  * data-parallel triplets + ONE all-reduce of dL/dE0 per step == the single-process step on the
    concatenated batch (the reference run with batch_size = world * B);
  * test users split across ranks + all-reduced metric sums == the single-process evaluation;
  * every rank's slice of the epoch stream is disjoint and their union is the epoch.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neurec_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _graph():
    from oracle import train
    rng = np.random.RandomState(0)
    U, I, d = 60, 50, 16
    users = np.repeat(np.arange(U), 5)
    items = np.concatenate([rng.choice(I, 5, replace=False) for _ in range(U)])
    A = train.lightgcn_adjacency(users, items, U, I, "pre")
    E0 = (rng.randn(U + I, d) * 0.1).astype(np.float64)
    return A.astype(np.float64), E0, U, I


def _worker(rank, world, port, out):
    from oracle import native, train
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    comm = parallel.init_from_env(backend="gloo")
    assert comm.active and comm.rank == rank and comm.world == world
    A, E0, U, I = _graph()
    m, v = np.zeros_like(E0), np.zeros_like(E0)
    adam = train.Adam(0.01, dtype=np.float64)
    rng = np.random.RandomState(5)
    B = 32
    for step in range(3):
        bu, bp, bn = rng.randint(0, U, world * B), rng.randint(0, I, world * B), rng.randint(0, I, world * B)
        lo, hi = parallel.partition(world * B, rank, world)                 # this rank's triplets
        _, _, g = train.lightgcn_loss_and_grad(A, A.T.tocsr(), E0, U, 2, bu[lo:hi], bp[lo:hi],
                                               bn[lo:hi], 1e-3)
        gt = torch.from_numpy(g)
        comm.allreduce_sum_(gt)                                             # the one exchange step
        adam.dense(E0, m, v, gt.numpy())
        adam.advance()
    # evaluation: users sharded, metric sums all-reduced
    users = np.arange(U, dtype=np.int32)
    mine = parallel.shard_users(users, rank, world)
    S = native.score_gemm(E0[:U].astype(np.float32), mine, E0[U:].astype(np.float32))
    truth = [[int((u * 7) % I)] for u in mine]
    sums = torch.from_numpy(native.eval_matrix(S, truth, [2, 4], 10).astype(np.float64).sum(0))
    comm.allreduce_sum_(sums)
    slowest = comm.max_float(float(rank))
    comm.barrier()
    if rank == 0:
        np.savez(out, E0=E0, means=(sums / U).numpy(), slowest=slowest)
    comm.shutdown()


def test_two_rank_training_and_eval_match_single_process(tmp_path):
    from oracle import native, train
    out = str(tmp_path / "rank0.npz")
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    # single process, batch = world * B
    A, E0, U, I = _graph()
    m, v = np.zeros_like(E0), np.zeros_like(E0)
    adam = train.Adam(0.01, dtype=np.float64)
    rng = np.random.RandomState(5)
    for step in range(3):
        bu, bp, bn = rng.randint(0, U, 64), rng.randint(0, I, 64), rng.randint(0, I, 64)
        train.lightgcn_step(A, A.T.tocsr(), E0, m, v, U, 2, bu, bp, bn, 1e-3, adam)
    assert np.abs(got["E0"] - E0).max() < 1e-12
    S = native.score_gemm(E0[:U].astype(np.float32), None, E0[U:].astype(np.float32))
    want = native.eval_matrix(S, [[int((u * 7) % I)] for u in range(U)], [2, 4], 10).astype(np.float64).mean(0)
    np.testing.assert_allclose(got["means"], want, rtol=1e-12)
    assert got["slowest"] == 1.0                                             # max over ranks


def _gather_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    comm = parallel.init_from_env(backend="gloo")
    mine = [torch.arange(5, dtype=torch.int32) + 100 * rank + 10 * k for k in range(3)]
    tok_a = comm.allgather_cat_start(mine)                     # two collectives in flight
    tok_b = comm.allgather_cat_start([t + 1000 for t in mine])
    a, b = comm.allgather_cat_finish(tok_a), comm.allgather_cat_finish(tok_b)
    bc = torch.full((3,), float(rank))
    comm.broadcast_(bc, 1)
    assert bc.tolist() == [1.0, 1.0, 1.0]
    if rank == 1:
        np.savez(out, a=torch.stack(a).numpy(), b=torch.stack(b).numpy())
    comm.barrier()
    comm.shutdown()


def test_triplet_allgather_is_rank_major_concatenation(tmp_path):
    out = str(tmp_path / "g.npz")
    mp.start_processes(_gather_worker, args=(2, _free_port(), out), nprocs=2, join=True,
                       start_method="spawn")
    got = np.load(out)
    want = np.stack([np.concatenate([np.arange(5) + 100 * r + 10 * k for r in range(2)]) for k in range(3)])
    np.testing.assert_array_equal(got["a"], want)
    np.testing.assert_array_equal(got["b"], want + 1000)
    # single process: the helper is the identity
    comm = parallel.Comm()
    parts = [torch.arange(3, dtype=torch.int32)] * 3
    assert comm.allgather_cat_finish(comm.allgather_cat_start(parts))[1] is parts[1]


def test_partition_covers_range_without_overlap():
    for n in (0, 1, 7, 1024, 813886):
        for world in (1, 2, 3, 8):
            cuts = [parallel.partition(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    users = np.arange(10)
    assert np.concatenate([parallel.shard_users(users, r, 3) for r in range(3)]).tolist() == list(range(10))


def test_single_process_comm_is_a_no_op():
    comm = parallel.Comm()
    t = torch.ones(3)
    assert comm.allreduce_sum_(t) is t and not comm.active and comm.max_float(2.5) == 2.5
    comm.barrier()


def _rows_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    comm = parallel.init_from_env(backend="gloo")
    b = parallel.block_size(7, world)                                   # 7 rows over 2 ranks: blocks of 4
    local = torch.full((b, 3), float(rank + 1))
    table = torch.zeros(b * world, 3)
    comm.all_gather_rows(local, table)
    # rank r sends r+1 rows to rank 0 and 2 rows to rank 1, payload = 10*rank + destination
    counts = [rank + 1, 2]
    send = torch.cat([torch.full((c, 2), 10.0 * rank + dst) for dst, c in enumerate(counts)])
    recv, rc = comm.all_to_all_rows(send, counts)
    back, bc = comm.all_to_all_rows(recv + 100.0, rc)                  # answers travel the reverse route
    np.savez(out % rank, table=table.numpy(), recv=recv.numpy(), rc=np.asarray(rc), back=back.numpy(),
             bc=np.asarray(bc))
    comm.barrier()
    comm.shutdown()


def test_row_sharding_collectives(tmp_path):
    """the two exchange primitives of the row-sharded engine (neurec_amd/sharded.py) on gloo"""
    out = str(tmp_path / "rows%d.npz")
    mp.start_processes(_rows_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(out % 0), np.load(out % 1)
    want_table = np.repeat(np.array([1.0, 2.0]), 4)[:, None] * np.ones((1, 3))
    np.testing.assert_array_equal(r0["table"], want_table)
    np.testing.assert_array_equal(r1["table"], want_table)
    assert r0["rc"].tolist() == [1, 2] and r1["rc"].tolist() == [2, 2]
    np.testing.assert_array_equal(r0["recv"][:, 0], [0, 10, 10])        # from rank 0 (1 row), rank 1 (2 rows)
    np.testing.assert_array_equal(r1["recv"][:, 0], [1, 1, 11, 11])
    assert r0["bc"].tolist() == [1, 2] and r1["bc"].tolist() == [2, 2]
    np.testing.assert_array_equal(r0["back"][:, 0], [100, 101, 101])     # my rows, answered, in my send order
    np.testing.assert_array_equal(r1["back"][:, 0], [110, 110, 111, 111])
    assert parallel.block_size(7, 2) == 4 and parallel.block_size(8, 2) == 4 and parallel.block_size(1, 8) == 1


def test_bipartite_partition_balances_users_and_items():
    """parallel.BipartitePartition: every rank owns a slice of the users AND of the items (a contiguous
    cut of [users; items] would give the last rank every item row — half of all non-zeros at config 4);
    position() is a bijection onto the gathered layout, owner/local agree between numpy and torch."""
    import numpy as np
    import torch
    from neurec_amd.parallel import BipartitePartition
    for U, I, world in ((10, 7, 3), (943, 1682, 2), (1000, 100, 8), (5, 3, 8)):
        p = BipartitePartition(U, I, world)
        nodes = np.arange(U + I)
        owner, local = p.owner_local(nodes)
        pos = p.position(nodes)
        assert len(set(pos.tolist())) == U + I and pos.max() < p.n_pad
        to, tl = p.owner_local(torch.arange(U + I))
        assert (to.numpy() == owner).all() and (tl.numpy() == local).all()
        for r in range(world):
            (ulo, uhi), (ilo, ihi) = p.users_of(r), p.items_of(r)
            mine = np.flatnonzero(owner == r)
            assert set(mine.tolist()) == set(range(ulo, uhi)) | set(range(U + ilo, U + ihi))
            assert (local[ulo:uhi] == np.arange(uhi - ulo)).all()
            assert (local[U + ilo:U + ihi] == p.bu + np.arange(ihi - ilo)).all()
        gu, gi = p.gathered_index()
        assert (gu.numpy() == pos[:U]).all() and (gi.numpy() == pos[U:]).all()
        # balance: no rank holds more than its share (+1 block rounding) of either side
        assert max(uhi - ulo for ulo, uhi in (p.users_of(r) for r in range(world))) <= p.bu


# ------------------------------------------------------------------ column-sharded tables (neurec_amd/colshard.py)
def _colshard_worker(rank, world, port, out):
    """The algebra of the column-sharded step with the CPU oracle as each rank's compute: every rank holds d/W
    columns for all nodes, steps on the SAME global batch; the one exchange is the all-gather of the per-triplet
    partial inner products, summed in rank order (parallel.Comm.all_gather_rows, as colshard.py does it)."""
    from oracle import train
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    comm = parallel.init_from_env(backend="gloo")
    A, E0, U, I = _graph()
    At = A.T.tocsr()
    d, L, reg = E0.shape[1], 2, 1e-3
    dl = d // world
    E = E0[:, rank * dl:(rank + 1) * dl].copy()                              # my columns
    m, v = np.zeros_like(E), np.zeros_like(E)
    adam = train.Adam(0.01, dtype=np.float64)
    rng = np.random.RandomState(5)
    losses = []
    for step in range(3):
        bu, bp, bn = rng.randint(0, U, 64), rng.randint(0, I, 64), rng.randint(0, I, 64)     # the GLOBAL batch
        Estar, _ = train.lightgcn_propagate(A, E, L)                                          # column-wise: no exchange
        iu, ii, ij = bu, U + bp, U + bn
        eu, ei, ej = Estar[iu], Estar[ii], Estar[ij]
        zu, zi, zj = E[iu], E[ii], E[ij]
        part = np.stack([np.sum(eu * ei, 1), np.sum(eu * ej, 1),
                         0.5 * (np.sum(zu * zu, 1) + np.sum(zi * zi, 1) + np.sum(zj * zj, 1))], 1)
        allp = torch.empty((world,) + part.shape, dtype=torch.float64)
        comm.all_gather_rows(torch.from_numpy(part), allp.view(world * part.shape[0], 3))      # the one exchange
        given = allp[0].numpy().copy()
        for r in range(1, world):
            given = given + allp[r].numpy()                                                   # rank order
        x = given[:, 0] - given[:, 1]
        lb, g = train.bpr_terms(x)
        losses.append((float(np.sum(lb)), float(reg * np.sum(given[:, 2]))))
        Gstar = np.zeros_like(E)
        np.add.at(Gstar, iu, g[:, None] * (ei - ej))
        np.add.at(Gstar, ii, g[:, None] * eu)
        np.add.at(Gstar, ij, -g[:, None] * eu)
        H = Gstar / (L + 1)
        G = H
        for _ in range(L):
            G = H + train.spmm_rowwise(At, G)                                                  # column-wise again
        R = np.zeros_like(E)
        np.add.at(R, iu, reg * zu)
        np.add.at(R, ii, reg * zi)
        np.add.at(R, ij, reg * zj)
        adam.dense(E, m, v, G + R)
        adam.advance()
    np.savez(out % rank, E=E, losses=np.asarray(losses))
    comm.shutdown()


def test_column_sharded_step_equals_single_process(tmp_path):
    from oracle import train
    out = str(tmp_path / "rank%d.npz")
    mp.start_processes(_colshard_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    r = [np.load(out % k) for k in range(2)]
    np.testing.assert_array_equal(r[0]["losses"], r[1]["losses"])           # both ranks: the same summed products
    A, E0, U, I = _graph()
    m, v = np.zeros_like(E0), np.zeros_like(E0)
    adam = train.Adam(0.01, dtype=np.float64)
    rng = np.random.RandomState(5)
    want = []
    for step in range(3):
        bu, bp, bn = rng.randint(0, U, 64), rng.randint(0, I, 64), rng.randint(0, I, 64)
        want.append(train.lightgcn_step(A, A.T.tocsr(), E0, m, v, U, 2, bu, bp, bn, 1e-3, adam))
    np.testing.assert_allclose(r[0]["losses"], np.asarray(want), rtol=1e-12)
    np.testing.assert_allclose(np.concatenate([r[0]["E"], r[1]["E"]], 1), E0, atol=1e-12)


def _hop_worker(rank, world, port, out):
    """The two r05 forms of the row-sharded hop, their algebra on CPU (scipy plays the SpMM kernels): the column-sliced
    all-gather (`all_gather_rows_start/finish` per slab) and the reduced exchange (item blocks all-gathered, per-rank
    partials of the item rows through `all_to_all_equal_start/finish`, added in rank order)."""
    import scipy.sparse as sp
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    comm = parallel.init_from_env(backend="gloo")
    A, E0, U, I = _graph()
    A = A.tocsr()
    d = E0.shape[1]
    part = parallel.BipartitePartition(U, I, world)
    bu, bi, b = part.bu, part.bi, part.b
    (ulo, uhi), (ilo, ihi) = part.users_of(rank), part.items_of(rank)

    def block(table):
        blk = np.zeros((b, table.shape[1]))
        blk[:uhi - ulo] = table[ulo:uhi]
        blk[bu:bu + ihi - ilo] = table[U + ilo:U + ihi]
        return blk
    X = np.random.RandomState(3).randn(U + I, d)
    local = torch.from_numpy(block(X))
    # this rank's rows of A, columns as positions in the rank-major gathered layout
    rows = sp.vstack([A[ulo:uhi], sp.csr_matrix((bu - (uhi - ulo), U + I)), A[U + ilo:U + ihi],
                      sp.csr_matrix((bi - (ihi - ilo), U + I))]).tocoo()
    pos = part.position(rows.col.astype(np.int64))
    A_blk = sp.csr_matrix((rows.data, (rows.row, pos)), shape=(b, part.n_pad))
    # --- column-sliced: slab s + 1 is gathered while slab s is multiplied
    S, w = 2, d // 2
    Y = np.zeros((b, d))
    bufs = [torch.zeros(part.n_pad, w, dtype=torch.float64) for _ in range(2)]
    toks = [comm.all_gather_rows_start(local[:, :w].contiguous(), bufs[0]), None]
    for s in range(S):
        if s + 1 < S:
            toks[(s + 1) % 2] = comm.all_gather_rows_start(local[:, (s + 1) * w:(s + 2) * w].contiguous(), bufs[(s + 1) % 2])
        comm.all_gather_rows_finish(toks[s % 2])
        Y[:, s * w:(s + 1) * w] = A_blk @ bufs[s % 2].numpy()
    # --- reduced exchange: items all-gathered, item rows as per-rank partials
    Z = torch.zeros(world * bi, d, dtype=torch.float64)
    tok = comm.all_gather_rows_start(local[bu:].contiguous(), Z)
    At = A.T.tocsr()                                          # (A is symmetric here; written for the general case)
    mine_t = At[ulo:uhi][:, U:].tocoo()                       # my user rows of A^T, item columns: (u, i) = A[i][u]
    Mp = sp.csr_matrix((mine_t.data, (mine_t.col, mine_t.row)), shape=(world * bi, b))
    P = torch.from_numpy(Mp @ local.numpy())
    comm.all_gather_rows_finish(tok)
    R = torch.zeros(world * bi, d, dtype=torch.float64)
    tok = comm.all_to_all_equal_start(P.contiguous(), R)
    mu = A[ulo:uhi][:, U:]                                    # my user rows x every item (id order == gathered order)
    Mu = sp.hstack([mu, sp.csr_matrix((uhi - ulo, world * bi - I))]).tocsr()
    Yr = np.zeros((b, d))
    Yr[:uhi - ulo] = Mu @ Z.numpy()
    comm.all_to_all_equal_finish(tok)
    acc = R.view(world, bi, d)[0].clone()
    for q in range(1, world):
        acc += R.view(world, bi, d)[q]                        # rank order
    Yr[bu:] = acc.numpy()
    want = block(A @ X)
    np.savez(out % rank, sliced=np.abs(Y - want).max(), reduce=np.abs(Yr - want).max(), calls=str(dict(comm.calls)))
    comm.barrier()
    comm.shutdown()


def test_row_sharded_hop_forms_two_ranks(tmp_path):
    out = str(tmp_path / "hop%d.npz")
    mp.start_processes(_hop_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    for r in range(2):
        got = np.load(out % r)
        assert got["sliced"] < 1e-12 and got["reduce"] < 1e-12, (r, got["sliced"], got["reduce"])
        assert "all_gather_async" in str(got["calls"]) and "all_to_all_equal" in str(got["calls"])


def _row_gather_worker(rank, world, store, out):
    """file-store rendezvous (no MASTER_PORT at all) + the per-user row gather of the sharded drop-in evaluation"""
    for k in ("MASTER_PORT",):
        os.environ.pop(k, None)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      NEUREC_DIST_INIT_FILE=store)
    comm = parallel.init_from_env(backend="gloo")
    assert comm.active and comm.world == world
    n = (5, 0, 3)[rank]                                                     # unequal shares, an empty one
    users = torch.arange(n, dtype=torch.int32) + 100 * rank
    rows = (torch.arange(n * 4, dtype=torch.float32).view(n, 4) + 1000 * rank)
    got_u, got_r = parallel.gather_rows_by_user(comm, users, rows, torch.device("cpu"))
    if rank == 0:
        np.savez(out, users=got_u.numpy(), rows=got_r.numpy())
    comm.barrier()
    comm.shutdown()


def test_file_store_rendezvous_and_row_gather_over_three_ranks(tmp_path):
    """bench.py's self-started ranks and `-m neurec_amd.main` under a launcher meet through parallel.init_from_env;
    NEUREC_DIST_INIT_FILE replaces MASTER_ADDR / MASTER_PORT by a file store (no port to lose).  The sharded
    drop-in evaluation gathers (user, metric row) pairs of unequal shares rank-major: every rank then means the same rows
    in the same order as one process does (uni_evaluator.py:150-151)."""
    out = str(tmp_path / "g.npz")
    mp.start_processes(_row_gather_worker, args=(3, str(tmp_path / "store"), out), nprocs=3, join=True, start_method="spawn")
    got = np.load(out)
    want_u = np.concatenate([np.arange(5), 200 + np.arange(3)]).astype(np.int32)
    want_r = np.concatenate([np.arange(20, dtype=np.float32).reshape(5, 4), 2000 + np.arange(12, dtype=np.float32).reshape(3, 4)])
    np.testing.assert_array_equal(got["users"], want_u)
    np.testing.assert_array_equal(got["rows"], want_r)
