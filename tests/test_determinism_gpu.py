"""Deterministic row-gradient aggregation.

TF sums the IndexedSlices of a gather's gradient that hit the same row with
unsorted_segment_sum: in batch order, the item table's positive-lookup slices before its
negative-lookup slices (MF.py:57-72, LightGCN.py:99-104).  The HIP heads reproduce that
order through a batch plan (sorted (row, position) keys), so
  * the plan equals a host sort of the same keys,
  * the dense row gradients equal an ordered host accumulation of the per-occurrence rows
    (np.add.at — what oracle/train.py does) BIT FOR BIT, duplicates and hub rows included,
  * two runs of N steps leave identical tables, with or without a precomputed plan."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _host_plan(users, items, third, batch, n_users):
    out = []
    n = len(users)
    for b0 in range(0, n, batch):
        e = min(b0 + batch, n)
        nb = e - b0
        rows = [users[b0:e].astype(np.uint64)]
        rows.append(items[b0:e].astype(np.uint64) + np.uint64(n_users))
        if third is not None:
            rows.append(third[b0:e].astype(np.uint64) + np.uint64(n_users))
        rows = np.concatenate(rows)
        pos = np.arange(len(rows), dtype=np.uint64)
        assert len(rows) == (2 if third is None else 3) * nb
        out.append(np.sort((rows << np.uint64(32)) | pos))
    return np.concatenate(out)


@pytest.mark.parametrize("n,batch", [(1, 1), (700, 256), (4096, 1024), (5000, 512), (9000, 4096),
                                     # beyond one workgroup's LDS (2 * batch > 16384): the segmented network
                                     (8193, 8193), (50000, 20000), (70001, 33000), (300000, 150000)])
def test_batch_plan_equals_host_sort(n, batch):
    from neurec_amd import engine as E
    rng = np.random.RandomState(n)
    U, I = 37, 53                                    # few rows: long runs of duplicates
    users = rng.randint(0, U, n).astype(np.int32)
    pos = rng.randint(0, I, n).astype(np.int32)
    neg = rng.randint(0, I, n).astype(np.int32)
    got3 = E.bpr_plan(_dev(users), _dev(pos), _dev(neg), batch, U).cpu().numpy().view(np.uint64)
    np.testing.assert_array_equal(got3, _host_plan(users, pos, neg, batch, U))
    got2 = E.bpr_plan(_dev(users), _dev(pos), None, batch, U).cpu().numpy().view(np.uint64)
    np.testing.assert_array_equal(got2, _host_plan(users, pos, None, batch, U))


@pytest.mark.parametrize("n", [16385, 40000, 65536, 65537, 1 << 20])
def test_key_sort_of_any_length(n):
    """nrhip_sort_u64 beyond 16384 keys (the row-sharded engines sort a global batch's returning rows)"""
    from neurec_amd import engine as E
    rng = np.random.RandomState(n % 1000)
    keys = (rng.randint(0, 5000, n).astype(np.int64) << 32) | rng.permutation(n).astype(np.int64)
    got = E.sort_keys(_dev(keys)).cpu().numpy()
    np.testing.assert_array_equal(got, np.sort(keys))


def test_many_segments_sorted_in_one_launch_and_epoch_routing_equals_per_batch_routing():
    """nrhip_sort_u64_segments (one workgroup per segment, lengths 0 .. 16384) and the epoch-at-once routing tables
    of the row-sharded engines (nrhip_route_epoch / nrhip_route_epoch_owner_keys) against the per-batch calls."""
    import torch
    from neurec_amd import engine as E, parallel
    from neurec_amd.sharded import RowRouter
    rng = np.random.RandomState(11)
    lens = np.array([0, 1, 2, 63, 64, 65, 129, 1000, 3072, 4097, 16384, 5], np.int32)
    off = np.concatenate([[0], np.cumsum(lens + 3)])[:-1].astype(np.int64)       # gaps between segments stay untouched
    total = int(off[-1] + lens[-1] + 3)
    keys = (rng.randint(0, 300, total).astype(np.int64) << 32) | rng.randint(0, 1 << 20, total).astype(np.int64)
    dk, doff, dlens = _dev(keys.copy()), _dev(off), _dev(lens)       # named: the call reads them after it returns
    E.call("nrhip_sort_u64_segments", E._ptr(dk), E._ptr(doff), E._ptr(dlens), len(lens), int(lens.max()), E._stream())
    want = keys.copy()
    for o, n in zip(off, lens):
        want[o:o + n] = np.sort(keys[o:o + n])
    np.testing.assert_array_equal(dk.cpu().numpy(), want)
    # routing tables of a whole stream (last batch short) == the per-batch routing, batch by batch
    U, I, B, n = 500, 700, 96, 96 * 5 + 17
    comm = parallel.Comm()
    part = parallel.BipartitePartition(U, I, 1)
    users, pos, neg = (rng.randint(0, hi, n).astype(np.int32) for hi in (U, I, I))
    du, dp, dn = _dev(users), _dev(pos), _dev(neg)
    planned, live = RowRouter(comm, part, B), RowRouter(comm, part, B)
    planned.plan_epoch([du, dp, dn], (0, U, U), B)
    live.plan_epoch([du, dp, dn], (0, U, U), B, tables=False)
    assert planned._tables is not None and live._tables is None
    for k in range((n + B - 1) // B):
        lo, hi = k * B, min(n, (k + 1) * B)
        a = planned.planned_route(k, hi - lo)
        b = live.request(du[lo:hi].contiguous(), dp[lo:hi].contiguous(), dn[lo:hi].contiguous(), U,
                         live.epoch_counts(k, hi - lo))
        for f in ("order", "inv", "asked", "asked_code"):
            assert torch.equal(getattr(a, f), getattr(b, f)), (k, f)
        ka, ia = planned.ordered_keys(a)
        kb, ib = live.ordered_keys(b)
        assert torch.equal(ka, kb) and torch.equal(ia[:3 * a.G], ib[:3 * b.G])


def test_steps_on_a_batch_beyond_the_lds_sort_match_the_oracle():
    """batch_size > 8192 used to fail in the default (lazy, fused) MF path.  One step of each
    engine on 20,000 triplets: plan sorted inside the step by the segmented network."""
    import torch
    from neurec_amd.trainer import LightGCNEngine, MFEngine
    from oracle import train as O
    rng = np.random.RandomState(3)
    U, I, d, B = 3000, 2000, 64, 20000
    users = rng.randint(0, U, B).astype(np.int32)
    pos = (rng.zipf(1.3, B) % I).astype(np.int32)                # hub items: runs of hundreds of occurrences
    neg = rng.randint(0, I, B).astype(np.int32)
    P0 = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q0 = (rng.randn(I, d) * 0.1).astype(np.float32)
    for kw in (dict(), dict(fused=False), dict(lazy=False)):
        mf = MFEngine(P0, Q0, 0.001, 0.01, B, **kw)
        loss = torch.zeros(2, device="cuda")
        mf.step(_dev(users), _dev(pos), _dev(neg), loss)
        P, Q = P0.copy(), Q0.copy()
        st = [np.zeros_like(x) for x in (P, P, Q, Q)]
        want = O.mf_step(P, Q, st[0], st[1], st[2], st[3], users, pos, neg, 0.01, O.Adam(0.001))
        assert abs(float(loss.sum()) - want) <= 1e-5 * want
        assert np.abs(mf.P.cpu().numpy() - P).max() <= 1e-5 and np.abs(mf.Q.cpu().numpy() - Q).max() <= 1e-5
    import scipy.sparse as sp
    R = sp.csr_matrix((np.ones(B, np.float32), (users, pos)), shape=(U, I))
    R.data[:] = 1
    coo = R.tocoo()
    A = O.lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
    E0 = (rng.randn(U + I, d) * 0.1).astype(np.float32)
    lg = LightGCNEngine(A, U, I, E0, 2, 0.01, 1e-3, B)
    loss = torch.zeros(2, device="cuda")
    lg.step(_dev(users), _dev(pos), _dev(neg), loss)
    e, m, v = E0.copy(), np.zeros_like(E0), np.zeros_like(E0)
    want = O.lightgcn_step(A, A, e, m, v, U, 2, users, pos, neg, 1e-3, O.Adam(0.01))
    A64 = A.astype(np.float64)
    e64, m64, v64 = E0.astype(np.float64), np.zeros(E0.shape), np.zeros(E0.shape)
    O.lightgcn_step(A64, A64, e64, m64, v64, U, 2, users, pos, neg, 1e-3, O.Adam(0.01, dtype=np.float64))
    got = loss.cpu().numpy()
    assert abs(got[0] - want[0]) <= 1e-5 * want[0] and abs(got[1] - want[1]) <= 1e-5 * want[1]
    # the first-moment estimate is 0.1 * g after one step: the gradient itself, at fp32 resolution
    # (item 1 is in ~6,000 of the 20,000 positives: its rows are 6,000-term fp32 sums — the head adds them in
    # batch order like np.add.at, the hops cut such a row into 64-nnz segments: the sums are reassociated)
    gscale = np.abs(m64).max()
    gm = lg.m.cpu().numpy()
    dm, dm32, dmo = np.abs(gm - m64).max(), np.abs(m - m64).max(), np.abs(gm - m).max()
    print("B = 20,000: first moment abs err vs fp64 %.1e of max %.1e (the fp32 restatement's own: %.1e); vs the fp32 "
          "restatement %.1e" % (dm, gscale, dm32, dmo))
    assert dmo <= 1e-6 * gscale and dm <= 1e-6 * gscale + dm32        # the same ordered 6,000-term sums
    # the table: 1e-5 of the fp64 twin wherever the gradient is resolved by fp32; where it is not (|g| within
    # 2^-17 of the largest: lr_t * m / (sqrt(v) + 1e-8) turns its rounding noise into a fraction of a step)
    # the fp32 restatement is just as far from its own twin — at most one step size
    resolved = np.abs(m64) >= gscale * 2.0 ** -17
    err = np.abs(lg.E0.cpu().numpy() - e64)
    assert err[resolved].max() <= 1e-5 + np.abs(e - e64)[resolved].max()
    assert err[~resolved].max() <= 0.01 and (~resolved).mean() < 0.5


def test_lazy_adam_steps_beyond_the_step_size_table():
    """the lazy MF step read lr_t from a table of 2^20 entries and refused step 2^20.  TF's lr_t
    is one value once both fp32 powers vanish against 1 (~17.3 k steps), so later steps are served from
    the table's tail.  Two engines — a table that ends at step 18,000 and the full one — run 18,300 steps
    (the batch loop in C) and must end bit-identical, in both lazy forms."""
    import torch
    from neurec_amd.trainer import MFEngine
    rng = np.random.RandomState(5)
    U, I, d, B, S = 300, 200, 64, 64, 18300
    users = _dev(rng.randint(0, U, S * B).astype(np.int32))
    pos = _dev(rng.randint(0, I, S * B).astype(np.int32))
    neg = _dev(rng.randint(0, I, S * B).astype(np.int32))
    P0 = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q0 = (rng.randn(I, d) * 0.1).astype(np.float32)
    for fused in (True, False):
        out = []
        for steps in (18000, 1 << 20):
            old = MFEngine.ALPHA_STEPS
            MFEngine.ALPHA_STEPS = steps
            try:
                mf = MFEngine(P0, Q0, 0.001, 0.0, B, fused=fused)
            finally:
                MFEngine.ALPHA_STEPS = old
            assert mf._alpha_tail_const
            losses = torch.zeros(2 * S, device="cuda")
            assert mf.run_batches(users, pos, neg, B, losses) == S
            mf.step(users[:B], pos[:B], neg[:B], losses[:2])            # the per-step entry too
            out.append((mf.P.cpu().numpy(), mf.Q.cpu().numpy(), mf.mP.cpu().numpy(), mf.vQ.cpu().numpy(),
                        losses.cpu().numpy()))
        for a, b in zip(*out):
            np.testing.assert_array_equal(a, b)
    # a table too short for its tail to be constant still refuses, by name
    old = MFEngine.ALPHA_STEPS
    MFEngine.ALPHA_STEPS = 1000
    try:
        mf = MFEngine(P0, Q0, 0.001, 0.0, B)
        losses = torch.zeros(2 * 1100, device="cuda")
        with pytest.raises(NotImplementedError, match="ALPHA_STEPS"):
            mf.run_batches(users[:1100 * B], pos[:1100 * B], neg[:1100 * B], B, losses)
    finally:
        MFEngine.ALPHA_STEPS = old


def _ordered_rows(n_rows, d, idx_lists, contrib_lists):
    out = np.zeros((n_rows, d), np.float32)
    for idx, c in zip(idx_lists, contrib_lists):
        np.add.at(out, idx, c)                       # sequential, in batch order
    return out


@pytest.mark.parametrize("d,U,I,B", [(64, 40, 30, 1024), (16, 300, 200, 512), (128, 5, 7, 333), (48, 64, 64, 64),
                                     (64, 3, 2, 1000), (64, 1, 1, 16), (64, 1, 1, 17), (32, 2, 1, 48)])
def test_mf_head_row_sums_are_the_ordered_sums_bit_for_bit(d, U, I, B):
    """Per-occurrence rows come from the kernel itself (a batch without duplicates is their
    ground truth: each row then holds exactly one occurrence), the duplicate-heavy batch must
    equal their np.add.at accumulation."""
    import torch
    from neurec_amd import engine as E
    rng = np.random.RandomState(d + B)
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    users = rng.randint(0, U, B).astype(np.int32)
    pos = rng.randint(0, I, B).astype(np.int32)
    neg = rng.randint(0, I, B).astype(np.int32)
    reg = 0.01
    # expanded problem: triplet b owns private copies of its three rows -> no duplicates at all
    Px, Qx = P[users], np.concatenate([Q[pos], Q[neg]])
    ar = np.arange(B, dtype=np.int32)
    GPx, GQx = torch.zeros(B, d, device="cuda"), torch.zeros(2 * B, d, device="cuda")
    work, l2 = torch.zeros(8 * B, device="cuda"), torch.zeros(2, device="cuda")
    E.bpr_mf_grad(_dev(Px), _dev(Qx), _dev(ar), _dev(ar), _dev(ar + B), reg, GPx, GQx, work, l2)
    occ_u, occ_q = GPx.cpu().numpy(), GQx.cpu().numpy()
    want_P = _ordered_rows(U, d, [users], [occ_u])
    want_Q = _ordered_rows(I, d, [pos, neg], [occ_q[:B], occ_q[B:]])
    # the real batch, twice: plan sorted inside the call / plan handed in
    for given in (False, True):
        GP, GQ = torch.zeros(U, d, device="cuda"), torch.zeros(I, d, device="cuda")
        l2b = torch.zeros(2, device="cuda")
        plan = E.bpr_plan(_dev(users), _dev(pos), _dev(neg), B, U) if given else None
        E.bpr_mf_grad(_dev(P), _dev(Q), _dev(users), _dev(pos), _dev(neg), reg, GP, GQ, work, l2b, plan)
        np.testing.assert_array_equal(GP.cpu().numpy(), want_P)
        np.testing.assert_array_equal(GQ.cpu().numpy(), want_Q)
        np.testing.assert_array_equal(l2b.cpu().numpy(), l2.cpu().numpy())   # same terms, same order


@pytest.mark.parametrize("U,I,B", [(23, 31, 777), (2, 3, 1500), (1, 1, 16), (1, 2, 33), (400, 300, 4096)])
def test_lightgcn_head_row_sums_are_the_ordered_sums_bit_for_bit(U, I, B):
    """Runs of one row longer than a workgroup's 16 occurrences (hub items of a large global batch)
    are continued by the whole workgroup: same ordered sums."""
    import torch
    from neurec_amd import engine as E
    rng = np.random.RandomState(9)
    d, L = 64, 3
    N = U + I
    Es = rng.randn(N, d).astype(np.float32)
    E0 = (rng.randn(N, d) * 0.1).astype(np.float32)
    users = rng.randint(0, U, B).astype(np.int32)
    pos = rng.randint(0, I, B).astype(np.int32)
    neg = rng.randint(0, I, B).astype(np.int32)
    node = np.concatenate([users, U + pos, U + neg])
    ar = np.arange(B, dtype=np.int32)
    work, l2 = torch.zeros(8 * B, device="cuda"), torch.zeros(2, device="cuda")
    Gx, Rx = torch.zeros(3 * B, d, device="cuda"), torch.zeros(3 * B, d, device="cuda")
    E.lightgcn_bpr_grad(_dev(Es[node]), _dev(E0[node]), B, L, _dev(ar), _dev(ar), _dev(ar + B), 1e-3,
                        Gx, Rx, work, l2)
    want_G = _ordered_rows(N, d, [node], [Gx.cpu().numpy()])
    want_R = _ordered_rows(N, d, [node], [Rx.cpu().numpy()])
    G, R = torch.zeros(N, d, device="cuda"), torch.zeros(N, d, device="cuda")
    l2b = torch.zeros(2, device="cuda")
    E.lightgcn_bpr_grad(_dev(Es), _dev(E0), U, L, _dev(users), _dev(pos), _dev(neg), 1e-3, G, R, work, l2b)
    np.testing.assert_array_equal(G.cpu().numpy(), want_G)
    np.testing.assert_array_equal(R.cpu().numpy(), want_R)
    np.testing.assert_array_equal(l2b.cpu().numpy(), l2.cpu().numpy())


def test_fifty_steps_twice_give_identical_tables():
    """LightGCN (gowalla-like duplicates: positives drawn by degree) and BPR-MF: two engines fed the
    same 50 batches end bit-identical; so does a run that hands the sampler's plans in."""
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.graph import lightgcn_adjacency
    from neurec_amd.trainer import BprEpochSampler, LightGCNEngine, MFEngine
    train, _ = synth.interactions("gowalla", seed=3, scale=0.05)
    U, I = train.shape
    coo = train.tocoo()
    A = lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
    E0 = synth.xavier_uniform(U + I, 64, np.random.RandomState(2017))
    trc = E.DeviceCSR.from_scipy(train)
    runs = []
    for given in (False, False, True):
        lg = LightGCNEngine(A, U, I, E0, 3, 0.01, 1e-3, 1024)
        rs = np.random.RandomState(1)
        mf = MFEngine((rs.randn(U, 64) * 0.01).astype(np.float32), (rs.randn(I, 64) * 0.01).astype(np.float32),
                      0.001, 0.01, 1024)
        sampler = BprEpochSampler(trc, I, batch_size=1024, seed=11, plan_users=U if given else None)
        loss = torch.zeros(2, device="cuda")
        n = 0
        while n < 50:
            for b in sampler.batches():
                assert (b.plan is not None) == given
                lg.step(b[0], b[1], b[2], None, plan=b.plan)
                mf.step(b[0], b[1], b[2], loss, plan=b.plan)
                n += 1
                if n == 50:
                    break
        runs.append((lg.E0.cpu().numpy(), mf.P.cpu().numpy(), mf.Q.cpu().numpy(), loss.cpu().numpy()))
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            np.testing.assert_array_equal(a, b)


def test_loss_reduction_by_the_last_block_sees_every_term():
    """finish_loss hands the per-triplet terms to the block that finishes last through agent-scope
    stores and a relaxed counter (csrc/bpr.hip): stress it — 300 launches, every reduced loss must
    equal the fp64-accumulated sum of the terms the launch left in the work buffer."""
    import torch
    from neurec_amd import engine as E
    rng = np.random.RandomState(4)
    U, I, d, B = 2000, 3000, 64, 2048
    P, Q = _dev((rng.randn(U, d) * 0.3).astype(np.float32)), _dev((rng.randn(I, d) * 0.3).astype(np.float32))
    GP, GQ = torch.zeros(U, d, device="cuda"), torch.zeros(I, d, device="cuda")
    work = torch.zeros(8 * B, device="cuda")
    outs = torch.zeros(300, 2, device="cuda")
    want = np.zeros((300, 2), np.float32)
    ids = [(_dev(rng.randint(0, U, B).astype(np.int32)), _dev(rng.randint(0, I, B).astype(np.int32)),
            _dev(rng.randint(0, I, B).astype(np.int32))) for _ in range(300)]
    terms = []
    for k, (u, p, n) in enumerate(ids):
        E.bpr_mf_grad(P, Q, u, p, n, 0.05, GP, GQ, work, outs[k])
        terms.append(work[:2 * B].clone())
    got = outs.cpu().numpy()
    for k, t in enumerate(terms):
        t = t.cpu().numpy().astype(np.float64)
        want[k] = np.float32(t[:B].sum()), np.float32(0.05) * np.float32(t[B:].sum())
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("fused", [True, False], ids=["one-launch", "two-launch"])
@pytest.mark.parametrize("d,period,reg", [(64, 64, 0.0), (64, 5, 0.01), (128, 1000, 0.01), (20, 1, 0.0)])
def test_lazy_sparse_adam_is_bit_identical_to_the_sweep(d, period, reg, fused):
    """nrhip_adam_sparse_tf_lazy (exact lazy replay) against nrhip_adam_sparse_tf (TF's
    literal all-rows update, the checker): after 230 steps + flush the tables AND both moments are
    bit-equal — rows touched every step, rows touched once, rows never touched, with a replay bound
    (period) shorter and longer than the run."""
    import torch
    from neurec_amd.trainer import MFEngine
    rng = np.random.RandomState(d + period)
    U, I, B = 700, 900, 256
    P0 = (rng.randn(U, d) * 0.05).astype(np.float32)
    Q0 = (rng.randn(I, d) * 0.05).astype(np.float32)
    lazy = MFEngine(P0, Q0, 0.003, reg, B, lazy=True, lazy_period=period, fused=fused)
    sweep = MFEngine(P0, Q0, 0.003, reg, B, lazy=False)
    la, lb = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    hot_u, hot_i = np.arange(40), np.arange(60)          # most traffic on a few rows, a tail touched rarely,
    for step in range(230):                              # users >= 600 / items >= 800 never
        pick = lambda hot, n: np.where(rng.rand(B) < 0.7, rng.choice(hot, B), rng.randint(0, n, B)).astype(np.int32)
        bu, bp, bn = _dev(pick(hot_u, 600)), _dev(pick(hot_i, 800)), _dev(pick(hot_i, 800))
        lazy.step(bu, bp, bn, la)
        sweep.step(bu, bp, bn, lb)
        if step % 50 == 7:                               # reading the tables mid-run flushes; training goes on
            np.testing.assert_array_equal(lazy.P.cpu().numpy(), sweep.P.cpu().numpy())
        assert float(la[0]) == float(lb[0]) and float(la[1]) == float(lb[1])
    for name in ("P", "Q", "mP", "mQ", "vP", "vQ"):
        np.testing.assert_array_equal(getattr(lazy, name).cpu().numpy(), getattr(sweep, name).cpu().numpy(), err_msg=name)
    np.testing.assert_array_equal(lazy.P.cpu().numpy()[600:], P0[600:])          # never touched: never moved
    if not fused:                                   # (the one-launch step keeps no gradient table)
        assert not lazy.GP.cpu().numpy().any() and not lazy.GQ.cpu().numpy().any()     # gradients re-armed


@pytest.mark.parametrize("fused", [True, False], ids=["one-launch", "two-launch"])
def test_lazy_adam_with_next_batch_plans_is_bit_identical_to_the_sweep(fused):
    """The sampler's batches carry their own plan and the next batch's: the optimiser launch of step t
    brings the rows step t+1 will gather up to date, the gradient kernel then replays nothing.  Across
    epoch boundaries (no next plan on an epoch's last batch, short last batches) the tables and moments
    still equal the all-rows sweep bit for bit."""
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.trainer import BprEpochSampler, MFEngine
    tr, _ = synth.interactions("ml-100k", seed=4)
    U, I = tr.shape
    trc = E.DeviceCSR.from_scipy(tr)
    rs = np.random.RandomState(3)
    P0, Q0 = (rs.randn(U, 64) * 0.05).astype(np.float32), (rs.randn(I, 64) * 0.05).astype(np.float32)
    lazy = MFEngine(P0, Q0, 0.002, 0.01, 2048, lazy=True, lazy_period=16, fused=fused)
    sweep = MFEngine(P0, Q0, 0.002, 0.01, 2048, lazy=False)
    sampler = BprEpochSampler(trc, I, batch_size=2048, seed=5, plan_users=U)     # 39 batches per epoch
    la, lb = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    n, with_next = 0, 0
    for epoch in range(3):
        for b in sampler.batches():
            with_next += b.next_plan is not None
            lazy.step(b[0], b[1], b[2], la, plan=b.plan, next_plan=b.next_plan)
            sweep.step(b[0], b[1], b[2], lb, plan=b.plan)
            n += 1
            assert float(la[0]) == float(lb[0]) and float(la[1]) == float(lb[1]), n
    assert n == 3 * len(sampler) and with_next == 3 * (len(sampler) - 1)
    for name in ("P", "Q", "mP", "mQ", "vP", "vQ"):
        np.testing.assert_array_equal(getattr(lazy, name).cpu().numpy(), getattr(sweep, name).cpu().numpy(), err_msg=name)


def test_global_batch_of_8192_sorts_its_plan_inside_the_step():
    """The 8-GPU id-exchange mode steps on a global batch of 8 x 1,024 triplets whose plan can only be
    sorted inside the step (16,384 item occurrences: the LDS sort's limit): same tables as with the
    plan handed in, and the plan equals the host sort."""
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.graph import lightgcn_adjacency
    from neurec_amd.trainer import LightGCNEngine
    tr, _ = synth.interactions("gowalla", seed=3, scale=0.1)
    U, I = tr.shape
    coo = tr.tocoo()
    A = lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
    E0 = synth.xavier_uniform(U + I, 64, np.random.RandomState(1))
    rng = np.random.RandomState(2)
    B = 8192
    pick = rng.randint(0, coo.nnz, B)
    bu, bp, bn = coo.row[pick].astype(np.int32), coo.col[pick].astype(np.int32), rng.randint(0, I, B).astype(np.int32)
    plan = E.bpr_plan(_dev(bu), _dev(bp), _dev(bn), B, U)
    np.testing.assert_array_equal(plan.cpu().numpy().view(np.uint64), _host_plan(bu, bp, bn, B, U))
    outs = []
    for given in (False, True):
        lg = LightGCNEngine(A, U, I, E0, 3, 0.01, 1e-3, B)
        loss = torch.zeros(2, device="cuda")
        for _ in range(2):
            lg.step(_dev(bu), _dev(bp), _dev(bn), loss, plan=plan if given else None)
        outs.append((lg.E0.cpu().numpy(), loss.cpu().numpy()))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("mode", ["one-launch", "two-launch", "sweep"])
def test_epoch_batch_loop_in_one_native_call_equals_the_per_step_loop(mode):
    """MFEngine.run_batches (nrhip_mf_steps: the batch loop of MF.train_model in C) against one step()
    call per batch on the same epoch streams: per-step losses, tables and moments bit for bit, across
    two epochs (short last batch, no next plan at the epoch's end)."""
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.trainer import BprEpochSampler, MFEngine
    tr, _ = synth.interactions("ml-100k", seed=4)
    U, I = tr.shape
    trc = E.DeviceCSR.from_scipy(tr)
    rs = np.random.RandomState(3)
    P0, Q0 = (rs.randn(U, 64) * 0.05).astype(np.float32), (rs.randn(I, 64) * 0.05).astype(np.float32)
    kw = dict(lazy=False) if mode == "sweep" else dict(lazy=True, fused=mode == "one-launch")
    a, b = MFEngine(P0, Q0, 0.002, 0.01, 1024, **kw), MFEngine(P0, Q0, 0.002, 0.01, 1024, **kw)
    sa = BprEpochSampler(trc, I, batch_size=1000, seed=5, plan_users=U)
    sb = BprEpochSampler(trc, I, batch_size=1000, seed=5, plan_users=U)
    n = len(sa)
    la, lb = torch.zeros(n, 2, device="cuda"), torch.zeros(n, 2, device="cuda")
    for epoch in range(2):
        for k, bt in enumerate(sa.batches()):
            a.step(bt[0], bt[1], bt[2], la[k], plan=bt.plan, next_plan=bt.next_plan)
        users, pos, neg, plans = sb.epoch_stream()
        assert users.numel() % 1000 != 0                 # the last batch is short
        assert b.run_batches(users, pos, neg, 1000, lb, plans) == n and b.adam.t == a.adam.t
        np.testing.assert_array_equal(la.cpu().numpy(), lb.cpu().numpy())
    for name in ("P", "Q", "mP", "mQ", "vP", "vQ"):
        np.testing.assert_array_equal(getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy(), err_msg=name)
