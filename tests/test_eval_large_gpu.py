"""The pruned evaluator at item counts beyond gowalla's 40,981 (BASELINE configs[3]: I = 10^6, d = 128).

Above 393,216 items the level-2 rescoring takes the PACKED tile buckets (csrc/eval_select.hip: tile_count / tile_fill,
LDS histogram windows) instead of the strided ones; the score slab for redone rows is no longer a whole batch wide.
Checked here against the reference's own evaluator (oracle/_ref: evaluate.h:23-72 compiled as it is, else the C
restatement) fed with the fp32 fmaf-chain scores of sampled users — per-user metric rows bit for bit — and, for every
user, against the fp32 tile search and the materialised path of the HIP side.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _workload(U, I, d, seed, popular=True):
    """tables with a few items most users rank high (hot tiles: one bucket holding a pair of almost every row), train
    rows that contain some of each third user's best items (strikes that change the top K), test rows disjoint"""
    import torch
    rng = np.random.RandomState(seed)
    P = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q = (rng.randn(I, d) * 0.1).astype(np.float32)
    if popular:
        mean = P.mean(0) + 0.02
        hot = rng.choice(I, 40, replace=False)
        Q[hot] += (0.25 * mean / np.linalg.norm(mean)).astype(np.float32)
        P += (0.5 * mean).astype(np.float32)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    best = torch.cat([(Pd[lo:lo + 512] @ Qd.T).topk(8, dim=1).indices for lo in range(0, U, 512)]).cpu().numpy()
    tr_lists, te_lists = [], []
    for u in range(U):
        tr = set(rng.randint(0, I, rng.randint(0, 30)).tolist())
        if u % 3 == 0:
            tr |= set(best[u, :rng.randint(1, 8)].tolist())
        if u == 7:
            tr |= set(range(I - 45, I))                      # the partial last tile, struck
        te = set(rng.randint(0, I, rng.randint(0, 9)).tolist()) - tr
        if u % 5 == 0:
            te |= set(best[u, 5:8].tolist()) - tr            # hits among the top K
        tr_lists.append(sorted(tr))
        te_lists.append(sorted(te))
    return P, Q, Pd, Qd, tr_lists, te_lists


def _csr(E, lists, n_cols):
    from oracle.native import lists_to_csr
    ptr, idx = lists_to_csr(lists)
    return E.DeviceCSR(ptr, idx[:max(int(ptr[-1]), 1)], n_cols)


def _reference_rows(P, Q, users, tr_lists, te_lists, mids, k):
    """the reference evaluator on fmaf-chain scores of `users` (uni_evaluator.py:132-147 + evaluate.h:53-72)"""
    from oracle import native, ref
    S = native.score_gemm(P, users, Q)
    for r, u in enumerate(users):
        S[r, tr_lists[u]] = -np.inf
    fn = ref.eval_matrix if ref.available() else native.eval_matrix
    return fn(S, [te_lists[u] for u in users], mids, k)


@pytest.mark.parametrize("I,d,U", [(500_003, 64, 2600), (500_003, 128, 2100), (1_000_000, 128, 1100), (1_000_000, 64, 1100)])
def test_pruned_evaluation_beyond_the_strided_buckets_equals_the_reference_evaluator(I, d, U):
    import torch
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    P, Q, Pd, Qd, tr_lists, te_lists = _workload(U, I, d, seed=I % 1000 + d)
    trc, tec = _csr(E, tr_lists, I), _csr(E, te_lists, I)
    users = np.asarray([u for u in range(U) if te_lists[u]], np.int32)
    ud = torch.from_numpy(users).cuda()
    mids, k = [1, 2, 3, 4, 5], 20
    # 2 * ceil(I / 64) tiles: more than the strided buckets' LDS histogram holds
    assert 2 * ((I + 63) // 64) > 12288
    fast = FullRankEvaluator(trc, tec, mids, k, batch_rows=1024)             # the default search: int8
    rows = np.asarray(fast.evaluate_factors(Pd, Qd, ud, per_user=True))
    assert fast.search_used == "int8"
    assert fast.n_flagged <= len(users) // 10                                # the bound certifies: not a redo-everything run
    # the reference's evaluator on sampled users, the fmaf-chain scores as its input: identical metric rows
    rng = np.random.RandomState(1)
    pick = np.sort(rng.choice(len(users), 48, replace=False))
    pick = np.union1d(pick, np.flatnonzero(users == 7))                      # (the user with the struck last tile)
    want = _reference_rows(P, Q, users[pick], tr_lists, te_lists, mids, k)
    np.testing.assert_array_equal(rows[pick], want)
    assert want[:, k:2 * k].max() > 0                                        # recall: the test items are found
    # every user: the exact (fp32) tile search gives the same rows, and so do the means in one call
    exact = FullRankEvaluator(trc, tec, mids, k, batch_rows=1024, search="fp32")
    np.testing.assert_array_equal(np.asarray(exact.evaluate_factors(Pd, Qd, ud, per_user=True)), rows)
    if d > 64:                                                               # ... and the bf16 search (config 4's form in r05)
        bf = FullRankEvaluator(trc, tec, mids, k, batch_rows=1024, search="bf16")
        np.testing.assert_array_equal(np.asarray(bf.evaluate_factors(Pd, Qd, ud, per_user=True)), rows)
        assert bf.search_used == "bf16"
    means = fast.evaluate_factors(Pd, Qd, ud)
    np.testing.assert_allclose(means, rows.astype(np.float64).sum(0) / len(users), rtol=0, atol=1e-12)


def test_rows_redone_at_a_million_items_use_a_bounded_slab():
    """near-duplicate items make certificates fail: the flagged rows are redone from full fp32 rows, a bounded number
    at a time (a [batch_rows][I] slab would be 4 GB here), and come out as the materialised path's"""
    import torch
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    I, d, U = 1_000_000, 32, 300
    P, Q, Pd, Qd, tr_lists, te_lists = _workload(U, I, d, seed=3, popular=False)
    rng = np.random.RandomState(4)
    base = Q[:40].copy() * 4.0                                               # long items: they lead most users' rankings
    for c in range(200):                                                     # 8,000 near-copies spread over the tiles
        at = rng.choice(I, 40, replace=False)
        Q[at] = base * (1.0 + rng.randn(40, 1).astype(np.float32) * 1e-7)
    Qd = torch.from_numpy(Q).cuda()
    trc, tec = _csr(E, tr_lists, I), _csr(E, te_lists, I)
    users = np.asarray([u for u in range(U) if te_lists[u]], np.int32)
    ud = torch.from_numpy(users).cuda()
    ev = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=1024)
    rows = np.asarray(ev.evaluate_factors(Pd, Qd, ud, per_user=True))
    assert ev.n_flagged > 0
    assert ev._scores[0].shape[0] * ev._scores[0].shape[1] * 4 <= (1 << 30) + 4 * ev._scores[0].shape[1]
    pick = np.arange(0, len(users), 7)
    want = _reference_rows(P, Q, users[pick], tr_lists, te_lists, [1, 2, 3, 4, 5], 20)
    np.testing.assert_array_equal(rows[pick], want)


_CHILD = r'''
import json, sys
import numpy as np, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from neurec_amd import engine as E
from neurec_amd.trainer import FullRankEvaluator
from test_eval_large_gpu import _workload, _csr
out = {}
for U, I, d, nan_rows in %r:
    P, Q, Pd, Qd, tr_lists, te_lists = _workload(U, I, d, seed=U + d)
    if nan_rows:
        P[::17] = np.nan
        Pd = torch.from_numpy(P).cuda()
    trc, tec = _csr(E, tr_lists, I), _csr(E, te_lists, I)
    users = torch.from_numpy(np.asarray([u for u in range(U) if te_lists[u]], np.int32)).cuda()
    for search in ("fp32", "bf16"):
        ev = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=384, search=search)
        rows = np.asarray(ev.evaluate_factors(Pd, Qd, users, per_user=True))
        out["%%d/%%d/%%d/%%s" %% (U, I, d, search)] = [rows.view(np.uint32).tolist(), int(ev.n_flagged)]
print(json.dumps(out))
'''

_CASES = [(700, 5000, 64, False), (500, 33000, 128, False), (400, 3000, 32, True), (300, 2200, 16, False)]


def _run(form):
    env = dict(os.environ, NEUREC_RESCORE_GROUPED=form)
    out = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, os.path.join(ROOT, "tests"), _CASES)], env=env,
                         capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_packed_tile_buckets_equal_the_strided_buckets_and_the_per_row_kernel():
    """NEUREC_RESCORE_GROUPED = 2 (packed buckets at every shape) / 1 (strided where they fit) / 0 (one wave per
    user): the same metric rows bit for bit and the same number of redone rows — hot tiles that every user picks,
    the partial last tile, NaN user rows (their tile ids are whatever the workspace held: cases of different shapes
    run in ONE process on purpose), several batches with a short last one."""
    packed, strided, per_row = _run("2"), _run("1"), _run("0")
    assert packed.keys() == strided.keys() == per_row.keys() and len(packed) == 2 * len(_CASES)
    for key in packed:
        assert packed[key][1] == strided[key][1] == per_row[key][1], key
        np.testing.assert_array_equal(np.asarray(packed[key][0]), np.asarray(strided[key][0]), err_msg=key)
        np.testing.assert_array_equal(np.asarray(packed[key][0]), np.asarray(per_row[key][0]), err_msg=key)


@pytest.mark.parametrize("I,d", [(5000, 64), (70_000, 128)])
def test_native_redo_of_any_row_reproduces_the_pruned_rows(I, d):
    """nrhip_eval_redo ranks a flagged row again from a full fp32 score row — the materialised path, exact whatever
    made the search give up.  So marking rows the search had certified must change nothing: the same per-user metric
    rows bit for bit (and the reference evaluator's on sampled users), the same column sums, the marked count reported.
    Slabs shorter than the marked count (several passes) included."""
    import torch
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    U = 900
    P, Q, Pd, Qd, tr_lists, te_lists = _workload(U, I, d, seed=I % 97)
    trc, tec = _csr(E, tr_lists, I), _csr(E, te_lists, I)
    users_np = np.asarray([u for u in range(U) if te_lists[u]], np.int32)
    users = torch.from_numpy(users_np).cuda()
    n = len(users_np)
    clean = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=512)
    want_rows = np.asarray(clean.evaluate_factors(Pd, Qd, users, per_user=True))
    want_sums = clean.evaluate_factors(Pd, Qd, users, column_sums=True)
    assert clean.n_flagged == 0
    for k, batch_rows in ((1, 512), (n // 7, 512), (n // 3, 64)):          # (64-row slabs: n // 3 rows take several)
        ev = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=batch_rows)
        marked = torch.from_numpy(np.random.RandomState(k).choice(n, k, replace=False)).cuda()
        read = ev._read_native_sums
        state = {"mark": True}

        def patched():
            both = read()
            if state["mark"]:                                  # behind the search: as if these certificates had failed
                ev._flags_buf[marked] = 2
                both[-2] += k
                both[-1] += k
                state["mark"] = False
            return both
        ev._read_native_sums = patched
        rows = np.asarray(ev.evaluate_factors(Pd, Qd, users, per_user=True))
        assert ev.n_flagged == k and ev.n_uncertified == k
        np.testing.assert_array_equal(rows.view(np.uint32), want_rows.view(np.uint32))
        state["mark"] = True
        sums = ev.evaluate_factors(Pd, Qd, users, column_sums=True)
        assert ev.n_flagged == k
        np.testing.assert_array_equal(sums, want_sums)
    pick = np.arange(0, n, 11)
    np.testing.assert_array_equal(want_rows[pick], _reference_rows(P, Q, users_np[pick], tr_lists, te_lists, [1, 2, 3, 4, 5], 20))


@pytest.mark.parametrize("I,d", [(40_981, 32), (9_000, 64)])
def test_zero_factor_rows_take_the_equal_maxima_path_of_the_first_selection(I, d):
    """a zero user row scores 0 everywhere: every tile maximum is equal, more candidates than a wave has lanes — the
    first selection (select_tiles_kernel) ranks them by repeated maximum extraction; the row is then flagged for its
    ties and redone, and comes out as the materialised path's, like its neighbours"""
    import torch
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    U = 260
    P, Q, Pd, Qd, tr_lists, te_lists = _workload(U, I, d, seed=5)
    P[::5] = 0.0
    Pd = torch.from_numpy(P).cuda()
    trc, tec = _csr(E, tr_lists, I), _csr(E, te_lists, I)
    users = torch.from_numpy(np.asarray([u for u in range(U) if te_lists[u]], np.int32)).cuda()
    got, flagged = {}, {}
    for search in ("int8", "bf16", "fp32"):
        ev = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, search=search)
        got[search] = np.asarray(ev.evaluate_factors(Pd, Qd, users, per_user=True))
        flagged[search] = ev.n_flagged
    want = np.asarray(FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], 20, batch_rows=128, pruned=False)
                      .evaluate_factors(Pd, Qd, users, per_user=True))
    for search in got:
        assert flagged[search] >= int((P[users.cpu().numpy()] == 0).all(1).sum()), search
        np.testing.assert_array_equal(got[search].view(np.uint32), want.view(np.uint32), err_msg=search)


@pytest.mark.parametrize("top_k", [1, 5, 31, 40, 62])
def test_pruned_evaluation_at_other_cut_offs_equals_the_materialised_path(top_k):
    """K = 20 is the configured cut-off (NeuRec.properties); the kernels are sized by it — rank_compact_kernel<4> up to
    32 rescored tiles, <8> beyond, the first selection in registers while K + extra + 2 <= 64, the streaming ring at
    K = 62 — so every size class is run here against the materialised path, per-user rows bit for bit, and against
    the reference evaluator on sampled users"""
    import torch
    from neurec_amd import engine as E
    from neurec_amd.trainer import FullRankEvaluator
    U, I, d = 420, 12_000, 64
    P, Q, Pd, Qd, tr_lists, te_lists = _workload(U, I, d, seed=top_k)
    trc, tec = _csr(E, tr_lists, I), _csr(E, te_lists, I)
    users_np = np.asarray([u for u in range(U) if te_lists[u]], np.int32)
    users = torch.from_numpy(users_np).cuda()
    want = np.asarray(FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], top_k, batch_rows=160, pruned=False)
                      .evaluate_factors(Pd, Qd, users, per_user=True))
    for search in ("int8", "bf16", "fp32"):
        ev = FullRankEvaluator(trc, tec, [1, 2, 3, 4, 5], top_k, batch_rows=160, search=search)
        got = np.asarray(ev.evaluate_factors(Pd, Qd, users, per_user=True))
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg="%s K=%d" % (search, top_k))
        sums = ev.evaluate_factors(Pd, Qd, users, column_sums=True)
        np.testing.assert_allclose(sums, want.astype(np.float64).sum(0), rtol=1e-12)
    pick = np.arange(0, len(users_np), 9)
    np.testing.assert_array_equal(want[pick], _reference_rows(P, Q, users_np[pick], tr_lists, te_lists, [1, 2, 3, 4, 5], top_k))
