"""The lane-group SpMM schedule deals rows to workgroups by cost (r05, csrc/spmm_blocked.hip) instead of cutting the row
range into contiguous runs: a workgroup's cost no longer depends on how the nodes are numbered.  Guarded here:
  * the oracle's ascending-column row sums, bit for bit on rows of <= 64 non-zeros, and the work-item kernel's results
    (SpmmCSR.lane_group = False), full pass and both masked hops, d = 16 / 32 / 64, on a graph whose items are numbered
    by DESCENDING degree (the worst case for contiguous runs — r01-r04's schedule, which left the product in r06: all
    hub rows first, then thousands of short rows);
  * the pass time on that numbering stays within 10 % of the time on a shuffled numbering (runs: 1.4-1.5x,
    profiles/r05_exp_entcost.txt).
LightGCN.py:132-149 (what a hop computes) is numbering-invariant; so is its cost now."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(order):
    """gowalla-shaped twin (the bench's workload); items relabelled by `order`: 'degree' (descending) or 'shuffled'"""
    from neurec_amd import synth
    from neurec_amd.graph import lightgcn_adjacency
    train, _ = synth.interactions_around_test(
        synth.load_test_split(os.path.join(ROOT, "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
    U, I = train.shape
    deg = np.asarray(train.sum(0)).ravel()
    new_of = np.empty(I, np.int64)
    if order == "degree":
        new_of[np.argsort(-deg, kind="stable")] = np.arange(I)
    else:
        new_of[np.random.RandomState(1).permutation(I)] = np.arange(I)
    coo = train.tocoo()
    return lightgcn_adjacency(coo.row, new_of[coo.col], U, I, "pre"), U, I


def _csr(A, U, lane_group=True):
    from neurec_amd import engine as E
    csr = E.SpmmCSR.from_scipy(A, split_row=U)
    csr.lane_group = lane_group
    return csr


@pytest.mark.parametrize("d", [16, 32, 64])
def test_dealt_rows_give_the_ascending_column_sums(d):
    import torch
    from oracle import train as O
    A, U, I = _graph("degree")
    N = U + I
    rng = np.random.RandomState(d)
    X = rng.randn(N, d).astype(np.float32)
    add, acc = rng.randn(N, d).astype(np.float32), rng.randn(N, d).astype(np.float32)
    wanted = (rng.rand(N) < 0.05).astype(np.uint8)
    nonzero = (rng.rand(N) < 0.05).astype(np.uint8)
    Xs = X * nonzero[:, None]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    outs = {}
    for lane_group in (True, False):
        csr = _csr(A, U, lane_group)
        if lane_group:
            assert csr.ensure_schedule(d)                                         # dealing always fits
        res = []
        Y, S = torch.empty(N, d, device="cuda"), torch.empty(N, d, device="cuda")
        csr.matmul(dev(X), out=Y, addend=dev(add), sum_in=dev(acc), sum_out=S)
        res += [Y.cpu().numpy(), S.cpu().numpy()]
        if lane_group or d >= 64:                                                 # (the work-item kernel masks rows at d >= 64 only)
            Y2 = torch.full((N, d), 7.0, device="cuda")
            csr.matmul(dev(X), out=Y2, y_row_wanted=dev(wanted))                  # row-masked hop: other rows untouched
            res.append(Y2.cpu().numpy())
            Y3 = torch.empty(N, d, device="cuda")
            csr.matmul(dev(Xs), out=Y3, addend=dev(add), x_row_nonzero=dev(nonzero))  # column-masked hop
            res.append(Y3.cpu().numpy())
        outs[lane_group] = res
    short = np.diff(A.indptr) <= 64                                               # strict ascending-column rows
    for a, b in zip(outs[True], outs[False]):                                     # (hub rows: segments of 64 vs 256)
        np.testing.assert_array_equal(a[short], b[short])
        assert np.abs(a - b).max() < 1e-5
    want = O.spmm_rowwise(A, X) + add
    np.testing.assert_array_equal(outs[True][0][short], want[short])
    assert np.abs(outs[True][0] - want).max() < 1e-5
    assert (outs[True][2][wanted == 0] == 7.0).all()
    # the masked hops against the oracle's row sums (they are what the work-item comparison cannot cover below d = 64)
    full = O.spmm_rowwise(A, X)
    np.testing.assert_array_equal(outs[True][2][(wanted == 1) & short], full[(wanted == 1) & short])
    assert np.abs(outs[True][2][wanted == 1] - full[wanted == 1]).max() < 1e-5
    want3 = O.spmm_rowwise(A, Xs) + add
    np.testing.assert_array_equal(outs[True][3][short], want3[short])
    assert np.abs(outs[True][3] - want3).max() < 1e-5


def test_pass_time_does_not_depend_on_the_numbering():
    import torch
    times = {}
    for order in ("degree", "shuffled"):
        A, U, I = _graph(order)
        X = torch.randn(U + I, 64, device="cuda")
        Y = torch.empty_like(X)
        csr = _csr(A, U)
        for _ in range(5):
            csr.matmul(X, out=Y)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            csr.matmul(X, out=Y)
        b.record()
        torch.cuda.synchronize()
        times[order] = a.elapsed_time(b) / 50 * 1e3
    print("SpMM pass, d = 64, us: " + ", ".join("%s ids: %.1f" % (o, t) for o, t in sorted(times.items())))
    assert times["degree"] <= 1.10 * times["shuffled"]
