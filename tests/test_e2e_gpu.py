"""END-TO-END parity on the HEADLINE configuration (north_star: "NDCG@10 equal to reference ± 1e-5" on
LightGCN-gowalla): train K = 50 and 200 optimiser steps of BASELINE configs[2] (the bench's own workload: the reference's
real gowalla.test split, the synthetic train twin around it, L = 3, d = 64, B = 1,024, adj "pre", Adam lr 0.01, reg 1e-3)
on the HIP engine AND on the CPU restatement of LightGCN.py:132-180 (oracle.train fp32 — the pinned one — and an fp64
twin) FROM THE SAME TRIPLET STREAM, then evaluate every table the way the reference pipeline would print it
(LightGCN.py:183-192 -> cpp/uni_evaluator.py:101-157 -> evaluate.h: np.matmul scores, train items masked, the reference's
own C++ evaluator from oracle/_ref where it travelled, NeuRec.properties:34-41: five metrics, top 20, all test users),
and the HIP tables also with the HIP evaluator.  Asserted / printed:
  * |NDCG@10(HIP tables, HIP evaluator) − NDCG@10(HIP tables, reference C++ on the same fmaf-chain scores)| = 0, all 100
    metric columns identical;
  * |NDCG@10(HIP) − NDCG@10(oracle fp32)| against 1e-5 AND against the oracle's own fp32-vs-fp64 distance (Adam moves a
    coordinate by ~lr·sign(g): rounding-level gradient differences become table differences that NO fp32 evaluation
    order avoids — the restatement's distance to its own fp64 twin is the resolution of the comparison);
  * how many users' top-20 sets differ between the three tables."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
METRICS, TOPK = [1, 2, 4, 3, 5], 20          # Precision, Recall, NDCG, MAP, MRR ids of cpp/uni_evaluator.py:14-15
NDCG10 = 2 * TOPK + 9                        # METRICS[2] = 4 = NDCG


def _reference_eval(eu, ei, train, test, users, chain, want_top=False):
    """the reference's evaluation of (eu, ei): scores by np.matmul (MF.py:120-122 / LightGCN.py:118-119) or by the
    k-ascending fmaf chain (oracle.native.score_gemm), train items masked, C++ evaluator, 4,096 users per call"""
    from oracle import native, ref
    fn = ref.eval_matrix if ref.available() else native.eval_matrix
    rows, tops = [], []
    ip = train.indptr.astype(np.int64)
    for lo in range(0, len(users), 4096):
        ub = users[lo:lo + 4096]
        S = native.score_gemm(eu, ub, ei, threads=32) if chain else \
            np.ascontiguousarray(np.matmul(eu[ub], ei.T), dtype=np.float32)
        native.mask_train(S, ub, ip, train.indices)
        truth = [test.indices[test.indptr[u]:test.indptr[u + 1]].tolist() for u in ub]
        rows.append(fn(S, truth, METRICS, TOPK, threads=32))
        if want_top:
            part = np.argpartition(-S, TOPK, axis=1)[:, :TOPK]
            tops.append(np.sort(part, axis=1))
    per_user = np.concatenate(rows)
    return per_user, (np.concatenate(tops) if want_top else None)


def test_lightgcn_gowalla_train_then_evaluate_matches_the_reference_pipeline():
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.graph import lightgcn_adjacency
    from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, LightGCNEngine
    from oracle import ref, train as O
    from oracle.train_torch import TorchLightGCN
    train, test = synth.interactions_around_test(
        synth.load_test_split(os.path.join(ROOT, "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
    U, I = train.shape
    coo = train.tocoo()
    A = lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
    E0 = synth.xavier_uniform(U + I, 64, np.random.RandomState(2017))
    L, B, lr, reg = 3, 1024, 0.01, 1e-3
    lg = LightGCNEngine(A, U, I, E0, L, lr, reg, B)
    trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
    sampler = BprEpochSampler(trc, I, batch_size=B, seed=2018, plan_users=U)
    marks = (50, 200)
    batches = []
    for b in sampler.batches():
        batches.append(b)
        if len(batches) == marks[-1]:
            break
    host = [tuple(t.cpu().numpy() for t in b) for b in batches]
    users = np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)
    assert len(users) == 29858
    ev = FullRankEvaluator(trc, tec, METRICS, TOPK)
    t_users = torch.from_numpy(users).cuda()

    # the CPU side: the pinned fp32 restatement (scipy, one thread) and the fp64 twin (torch-CPU, threaded)
    A32 = A.astype(np.float32)
    e32, m32, v32 = E0.copy(), np.zeros_like(E0), np.zeros_like(E0)
    adam32 = O.Adam(lr, dtype=np.float32)
    twin = TorchLightGCN(A, E0, U, L, lr, reg, threads=32, dtype=np.float64)

    def cpu_tables(e):
        """E* = mean of the layers (LightGCN.py:142-149), as predict() uses it"""
        acc, ego = e.astype(np.float64), e.astype(np.float64)
        A64 = A.astype(np.float64)
        for _ in range(L):
            ego = A64 @ ego
            acc = acc + ego
        es = (acc / (L + 1)).astype(np.float32)
        return es[:U], es[U:]

    done = 0
    for K in marks:
        for k in range(done, K):
            lg.step(batches[k][0], batches[k][1], batches[k][2], None, plan=batches[k].plan)
            O.lightgcn_step(A32, A32, e32, m32, v32, U, L, host[k][0], host[k][1], host[k][2], reg, adam32)
            twin.step(*host[k])
        done = K
        e64 = twin.E.numpy()
        got_E = lg.E0.cpu().numpy()
        d32, d64, bar = np.abs(got_E - e32).max(), np.abs(got_E - e64).max(), np.abs(e32 - e64).max()
        # --- evaluation of the three tables by the reference pipeline, of the HIP tables also by the HIP evaluator
        eu_t, ei_t = lg.final_embeddings()
        eu, ei = eu_t.cpu().numpy(), ei_t.cpu().numpy()
        hip_rows = ev.evaluate_factors(eu_t.contiguous(), ei_t.contiguous(), t_users, per_user=True)
        ref_chain, _ = _reference_eval(eu, ei, train, test, users, chain=True)
        assert np.array_equal(hip_rows, ref_chain)                           # all 100 columns, all 29,858 users
        ref_hip, top_hip = _reference_eval(eu, ei, train, test, users, chain=False, want_top=True)
        ref_32, top_32 = _reference_eval(*cpu_tables(e32), train, test, users, chain=False, want_top=True)
        ref_64, top_64 = _reference_eval(*cpu_tables(e64), train, test, users, chain=False, want_top=True)
        nd = {k: float(np.mean(v.astype(np.float64), axis=0)[NDCG10]) for k, v in
              (("hip/hip", hip_rows), ("hip/ref-chain", ref_chain), ("hip/ref-matmul", ref_hip), ("cpu32", ref_32),
               ("cpu64", ref_64))}
        differ = lambda a, b: int((a != b).any(axis=1).sum())
        print("config 3 end to end, %d steps (%s evaluator): NDCG@10 HIP tables %.8f (HIP evaluator) / %.8f (reference "
              "C++, np.matmul scores) | oracle fp32 tables %.8f | fp64 twin %.8f ;  |HIP - fp32 oracle| = %.2e, oracle "
              "fp32-vs-fp64 = %.2e ;  E0 max abs diff: HIP vs fp32 oracle %.2e, HIP vs fp64 %.2e, fp32 oracle vs fp64 "
              "%.2e ;  users whose top-20 set differs: HIP vs fp32 oracle %d, fp32 oracle vs fp64 %d, HIP fmaf-chain vs "
              "np.matmul scores %d of %d"
              % (K, "the reference's own C++" if ref.available() else "the oracle's C++", nd["hip/hip"],
                 nd["hip/ref-matmul"], nd["cpu32"], nd["cpu64"], abs(nd["hip/hip"] - nd["cpu32"]),
                 abs(nd["cpu32"] - nd["cpu64"]), d32, d64, bar, differ(top_hip, top_32), differ(top_32, top_64),
                 int((hip_rows != ref_hip).any(axis=1).sum()), len(users)))
        assert nd["hip/hip"] == nd["hip/ref-chain"]
        assert abs(nd["hip/hip"] - nd["hip/ref-matmul"]) <= 1e-5             # BLAS order vs the fmaf chain: near-ties only
        # north_star's 1e-5, or the resolution of ANY fp32 run of these K steps (the restatement against its own twin)
        assert abs(nd["hip/hip"] - nd["cpu32"]) <= max(1e-5, 2.0 * abs(nd["cpu32"] - nd["cpu64"]))
        assert d32 <= max(1e-5, 2.0 * bar)
    assert nd["hip/hip"] > 5e-3                                                # the model learned something to rank


def test_bprmf_gowalla_train_then_evaluate_matches_the_reference_pipeline():
    """BASELINE configs[1] end to end (BPR-MF on gowalla, d = 64, B = 512, conf/MF.properties: lr 0.001, reg 0): 300
    steps of MF.py:85-113 on the HIP engine — the epoch's batch loop in one native call, one-launch lazy TF-sparse-Adam
    steps — and on oracle.train.mf_step (fp32, the pinned restatement with TF's all-rows sparse update, and an fp64
    twin) from the same triplet stream; then MF.predict's scores (MF.py:115-128) through the reference's own C++
    evaluator for all test users, the HIP tables also through the HIP evaluator."""
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, MFEngine
    from oracle import ref, train as O
    train, test = synth.interactions_around_test(
        synth.load_test_split(os.path.join(ROOT, "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
    U, I = train.shape
    d, B, lr, reg, K = 64, 512, 0.001, 0.0, 300
    rs = np.random.RandomState(2017)
    P0 = (rs.randn(U, d) * 0.01).astype(np.float32)
    Q0 = (rs.randn(I, d) * 0.01).astype(np.float32)
    trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
    mf = MFEngine(P0, Q0, lr, reg, B)
    sampler = BprEpochSampler(trc, I, neg_num=1, batch_size=B, shuffle=True, seed=2018, plan_users=U)
    mu, mp_, mn, plans = sampler.epoch_stream()
    n = K * B
    losses = torch.zeros(K, 2, device="cuda")
    mf.run_batches(mu[:n], mp_[:n], mn[:n], B, losses, plans[:3 * n])
    hu, hp, hn = (t[:n].cpu().numpy() for t in (mu, mp_, mn))

    def run(dt):
        P, Q = P0.astype(dt), Q0.astype(dt)
        mP, vP, mQ, vQ = (np.zeros(a.shape, dt) for a in (P, P, Q, Q))
        adam = O.Adam(lr, dtype=dt)
        out = [O.mf_step(P, Q, mP, vP, mQ, vQ, hu[k * B:(k + 1) * B], hp[k * B:(k + 1) * B], hn[k * B:(k + 1) * B],
                         reg, adam) for k in range(K)]
        return np.asarray(out, np.float64), P, Q
    l32, P32, Q32 = run(np.float32)
    l64, P64, Q64 = run(np.float64)
    got_l = losses.cpu().numpy().astype(np.float64).sum(1)
    gP, gQ = mf.P.cpu().numpy(), mf.Q.cpu().numpy()
    d32 = max(np.abs(gP - P32).max(), np.abs(gQ - Q32).max())
    bar = max(np.abs(P32 - P64).max(), np.abs(Q32 - Q64).max())
    users = np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)
    ev = FullRankEvaluator(trc, tec, METRICS, TOPK)
    hip_rows = ev.evaluate_factors(mf.P, mf.Q, torch.from_numpy(users).cuda(), per_user=True)
    ref_chain, _ = _reference_eval(gP, gQ, train, test, users, chain=True)
    assert np.array_equal(hip_rows, ref_chain)
    nd = {}
    for name, (p, q) in (("hip", (gP, gQ)), ("cpu32", (P32, Q32)), ("cpu64", (P64.astype(np.float32), Q64.astype(np.float32)))):
        rows, _ = _reference_eval(p, q, train, test, users, chain=False)
        nd[name] = float(np.mean(rows.astype(np.float64), axis=0)[NDCG10])
    nd["hip/hip"] = float(np.mean(hip_rows.astype(np.float64), axis=0)[NDCG10])
    print("config 2 end to end, %d steps (%s evaluator): epoch-loss rel err vs fp32 oracle %.1e; tables max abs diff HIP vs "
          "fp32 oracle %.2e (oracle fp32-vs-fp64 %.2e); NDCG@10 HIP tables %.8f (HIP evaluator) / %.8f (reference C++, "
          "np.matmul) | fp32 oracle %.8f | fp64 twin %.8f"
          % (K, "the reference's own C++" if ref.available() else "the oracle's C++",
             abs(got_l.sum() - l32.sum()) / abs(l32.sum()), d32, bar, nd["hip/hip"], nd["hip"], nd["cpu32"], nd["cpu64"]))
    assert abs(got_l.sum() - l32.sum()) <= 1e-5 * abs(l32.sum())                  # MF.py:110: the epoch's logged loss
    assert d32 <= max(1e-5, 2.0 * bar)
    assert abs(nd["hip/hip"] - nd["hip"]) <= 1e-5
    assert abs(nd["hip/hip"] - nd["cpu32"]) <= max(1e-5, 2.0 * abs(nd["cpu32"] - nd["cpu64"]))


def test_ngcf_gowalla_train_then_evaluate_matches_the_restated_pipeline():
    """BASELINE configs[4], NGCF half, end to end at the gowalla shape (d = 16, layers [16, 16], B = 512, lr 0.001,
    mess_dropout 0.1 — conf/NGCF.properties): 40 steps on the HIP engine and on oracle.train (fp32 and fp64; pinned to
    the reference's NGCF class on the small fixtures — the class itself densifies the 29,858 x 40,981 train matrix,
    NGCF.py:40, and cannot run at this size), dropout masks carried as data, then the evaluation forward (its own
    dropout draw, NGCF.py:140-141) and NGCF.predict's scores through the reference's C++ evaluator."""
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.graph import ngcf_adjacency, transpose_csr
    from neurec_amd.trainer import FullRankEvaluator, NGCFEngine
    from oracle import ref, train as O
    train, test = synth.interactions_around_test(
        synth.load_test_split(os.path.join(ROOT, "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
    U, I = train.shape
    A = ngcf_adjacency(train, "norm")
    At = transpose_csr(A)
    rng = np.random.RandomState(2017)
    d, B, lr, reg, drop, K = 16, 512, 0.001, 0.0, 0.1, 40
    E0 = synth.xavier_uniform(U + I, d, rng)
    W = [tuple((rng.randn(*s) * np.sqrt(1.3 * 2 / (s[0] + s[1]))).astype(np.float32)
               for s in ((d, d), (1, d), (d, d), (1, d))) for _ in range(2)]
    eng = NGCFEngine(A, At, U, I, E0, W, lr, reg, drop, B)
    coo = train.tocoo()
    steps = []
    for _ in range(K):
        pick = rng.randint(0, coo.nnz, B)
        steps.append(((coo.row[pick].astype(np.int32), coo.col[pick].astype(np.int32), rng.randint(0, I, B).astype(np.int32)),
                      [(rng.rand(U + I, d) < 1 - drop).astype(np.uint8) for _ in W]))
    eval_masks = [(rng.rand(U + I, d) < 1 - drop).astype(np.uint8) for _ in W]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    loss2 = torch.zeros(2, device="cuda")
    for (bu, bp, bn), masks in steps:
        eng.step(dev(bu), dev(bp), dev(bn), loss2, masks=[dev(m) for m in masks])
    out = eng.forward([dev(m) for m in eval_masks])
    eu_t, ei_t = out[:U].contiguous(), out[U:].contiguous()

    def run(dt):
        A_, At_ = A.astype(dt), At.astype(dt)
        oE = E0.astype(dt)
        oW = [[w.astype(dt) for w in ws] for ws in W]
        params = [oE] + [w for ws in oW for w in ws]
        ms, vs = [np.zeros_like(p) for p in params], [np.zeros_like(p) for p in params]
        ad = O.Adam(lr, dtype=dt)
        for (bu, bp, bn), masks in steps:
            _, dE, wg = O.ngcf_loss_and_grads(A_, At_, oE, [tuple(ws) for ws in oW], [m.astype(dt) for m in masks],
                                              1 - drop, U, bu, bp, bn, reg)
            for p, m, v, gg in zip(params, ms, vs, [dE] + [x for gs in wg for x in gs]):
                ad.dense(p, m, v, gg.reshape(p.shape))
            ad.advance()
        o, _ = O.ngcf_forward(A_, oE, [tuple(ws) for ws in oW], [m.astype(dt) for m in eval_masks], 1 - drop)
        return o.astype(np.float32)
    o32, o64 = run(np.float32), run(np.float64)
    users = np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)
    trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
    ev = FullRankEvaluator(trc, tec, METRICS, TOPK)
    hip_rows = ev.evaluate_factors(eu_t, ei_t, torch.from_numpy(users).cuda(), per_user=True)
    eu, ei = eu_t.cpu().numpy(), ei_t.cpu().numpy()
    ref_chain, _ = _reference_eval(eu, ei, train, test, users, chain=True)
    assert np.array_equal(hip_rows, ref_chain)
    nd = {"hip/hip": float(np.mean(hip_rows.astype(np.float64), axis=0)[NDCG10])}
    for name, o in (("hip", np.concatenate([eu, ei])), ("cpu32", o32), ("cpu64", o64)):
        rows, _ = _reference_eval(np.ascontiguousarray(o[:U]), np.ascontiguousarray(o[U:]), train, test, users, chain=False)
        nd[name] = float(np.mean(rows.astype(np.float64), axis=0)[NDCG10])
    d32, bar = np.abs(np.concatenate([eu, ei]) - o32).max(), np.abs(o32 - o64).max()
    print("config 5 (NGCF) end to end, %d steps (%s evaluator): evaluation embeddings max abs diff HIP vs fp32 oracle %.2e "
          "(oracle fp32-vs-fp64 %.2e); NDCG@10 HIP tables %.8f (HIP evaluator) / %.8f (reference C++, np.matmul) | fp32 "
          "oracle %.8f | fp64 twin %.8f" % (K, "the reference's own C++" if ref.available() else "the oracle's C++", d32, bar,
                                            nd["hip/hip"], nd["hip"], nd["cpu32"], nd["cpu64"]))
    # the embeddings: TF's Adam moves a coordinate whose gradient is at fp32 rounding level by up to a step size whichever
    # way the rounding falls (tests/test_tfgraph_gpu.py::test_ngcf_config5_three_steps_at_gowalla_size spells this out):
    # 40 steps of lr = 1e-3 leave a handful of such coordinates a few 1e-5 apart in ANY two fp32 runs — the restatement is
    # `bar` from its own fp64 twin — so the table bar is 5 x that; what the pipeline PRINTS, NDCG@10, is held to 1e-5
    assert d32 <= max(1e-5, 5.0 * bar)
    assert abs(nd["hip/hip"] - nd["hip"]) <= 1e-5
    assert abs(nd["hip/hip"] - nd["cpu32"]) <= max(1e-5, 2.0 * abs(nd["cpu32"] - nd["cpu64"]))


def test_multivae_gowalla_train_then_evaluate_matches_the_restated_pipeline():
    """BASELINE configs[4], Mult-VAE half, end to end at the gowalla shape (p_dim [16, 32], B = 512, lr 0.001, reg 0,
    tanh; conf/MultiVAE.properties at the narrow widths): 20 steps on the HIP engine (one native call per step, the
    decoder without a [B][I] logits buffer) and on oracle.train.multivae_loss_and_grads (fp32 / fp64; pinned to the
    reference's MultiVAE class, also at I = 40,981: tfgraph_big_multivae.npz), dropout masks and noise as data; then every
    test user scored on their own history at is_training = 0 (this package's documented default for predict) — logits as
    inner products of [g1(u) | 1] and [W_p1 | b_p1] — through the reference's C++ evaluator."""
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.trainer import FullRankEvaluator, MultiVAEEngine
    from oracle import ref, train as O
    train, test = synth.interactions_around_test(
        synth.load_test_split(os.path.join(ROOT, "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
    train.data[:] = 1.0
    U, I = train.shape
    z, h, B, lr, keep, K = 16, 32, 512, 0.001, 0.5, 20
    rs = np.random.RandomState(2017)
    s_ = 0.05
    params = {"Wq0": (rs.randn(I, h) * s_).astype(np.float32), "bq0": (rs.randn(h) * 0.01).astype(np.float32),
              "Wq1": (rs.randn(h, 2 * z) * s_).astype(np.float32), "bq1": (rs.randn(2 * z) * 0.01).astype(np.float32),
              "Wp0": (rs.randn(z, h) * s_).astype(np.float32), "bp0": (rs.randn(h) * 0.01).astype(np.float32),
              "Wp1t": (rs.randn(I, h) * s_).astype(np.float32), "bp1": (rs.randn(I) * 0.01).astype(np.float32)}
    trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
    eng = MultiVAEEngine(trc, I, params, lr, 0.0, "tanh", B)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    steps = []
    for k in range(K):
        rows = rs.choice(U, B, replace=False).astype(np.int32)
        drop_pos = (rs.rand(train.nnz) < keep).astype(np.float32)
        eps = (rs.randn(B, z) * 0.01).astype(np.float32)
        steps.append((rows, drop_pos, eps, min(0.2, k / 200.0)))
    for rows, drop_pos, eps, anneal in steps:
        eng.step(dev(rows), anneal, keep, drop_given=dev(drop_pos), eps_given=dev(eps), want_loss=False)
    pf, qf = eng.eval_factors()
    users = np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)
    ev = FullRankEvaluator(trc, tec, METRICS, TOPK)
    hip_rows = ev.evaluate_factors(pf, qf, torch.from_numpy(users).cuda(), per_user=True)

    def run(dt):
        p = {k: v.astype(dt) for k, v in params.items()}
        Wq, bq, Wp, bp = [p["Wq0"], p["Wq1"]], [p["bq0"], p["bq1"]], [p["Wp0"], p["Wp1t"].T.copy()], [p["bp0"], p["bp1"]]
        flat = Wq + bq + Wp + bp
        ms, vs = [np.zeros_like(x) for x in flat], [np.zeros_like(x) for x in flat]
        ad = O.Adam(lr, dtype=dt)
        for rows, drop_pos, eps, anneal in steps:
            X = np.asarray(train[rows].todense(), dtype=dt)
            D = np.ones_like(X)
            for b, u in enumerate(rows):
                lo, hi = train.indptr[u], train.indptr[u + 1]
                D[b, train.indices[lo:hi]] = drop_pos[lo:hi]
            _, (gWq, gbq, gWp, gbp), _ = O.multivae_loss_and_grads(X, Wq, bq, Wp, bp, D, keep, eps.astype(dt), anneal, 0.0, "tanh")
            for x, m, v, g in zip(flat, ms, vs, list(gWq) + list(gbq) + list(gWp) + list(gbp)):
                ad.dense(x, m, v, g.astype(dt).reshape(x.shape))
            ad.advance()
        # evaluation factors: g1(u) at is_training = 0 for every user, chunk by chunk
        G1 = np.zeros((U, h), np.float32)
        for lo in range(0, U, 2048):
            X = np.asarray(train[lo:lo + 2048].todense(), dtype=dt)
            _, _, _, cache = O.multivae_forward(X, Wq, bq, Wp, bp, np.ones_like(X), 1.0, np.zeros((X.shape[0], z), dt), 0.0, "tanh")
            G1[lo:lo + 2048] = cache[-1].astype(np.float32)
        return (np.concatenate([G1, np.ones((U, 1), np.float32)], 1),
                np.concatenate([Wp[1].T.astype(np.float32), bp[1].astype(np.float32)[:, None]], 1))
    f32, f64 = run(np.float32), run(np.float64)
    gp, gq = pf.cpu().numpy(), qf.cpu().numpy()
    ref_chain, _ = _reference_eval(gp, gq, train, test, users, chain=True)
    assert np.array_equal(hip_rows, ref_chain)
    nd = {"hip/hip": float(np.mean(hip_rows.astype(np.float64), axis=0)[NDCG10])}
    for name, (p_, q_) in (("hip", (gp, gq)), ("cpu32", f32), ("cpu64", f64)):
        rows_, _ = _reference_eval(np.ascontiguousarray(p_), np.ascontiguousarray(q_), train, test, users, chain=False)
        nd[name] = float(np.mean(rows_.astype(np.float64), axis=0)[NDCG10])
    d32 = max(np.abs(gp - f32[0]).max(), np.abs(gq - f32[1]).max())
    bar = max(np.abs(f32[0] - f64[0]).max(), np.abs(f32[1] - f64[1]).max())
    print("config 5 (Mult-VAE) end to end, %d steps (%s evaluator): evaluation factors max abs diff HIP vs fp32 oracle %.2e "
          "(oracle fp32-vs-fp64 %.2e); NDCG@10 HIP %.8f (HIP evaluator) / %.8f (reference C++, np.matmul) | fp32 oracle %.8f "
          "| fp64 twin %.8f" % (K, "the reference's own C++" if ref.available() else "the oracle's C++", d32, bar,
                                nd["hip/hip"], nd["hip"], nd["cpu32"], nd["cpu64"]))
    assert d32 <= max(1e-5, 5.0 * bar)
    assert abs(nd["hip/hip"] - nd["hip"]) <= 1e-5
    assert abs(nd["hip/hip"] - nd["cpu32"]) <= max(1e-5, 2.0 * abs(nd["cpu32"] - nd["cpu64"]))
