"""MF with every loss / optimiser of util/learner.py (MF.py:62-76): gradient kernels and the
TF-sparse row optimisers against oracle.train, five steps each, fp32 oracle for the update
arithmetic and fp64 twin for the loss."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


CASES = [(True, "bpr", "gd"), (True, "hinge", "adagrad"), (True, "square", "rmsprop"),
         (True, "bpr", "momentum"), (False, "cross_entropy", "adam"), (False, "square", "gd"),
         (False, "cross_entropy", "adagrad"), (True, "hinge", "adam")]


@pytest.mark.parametrize("pairwise,loss,learner", CASES)
def test_mf_variant_tracks_oracle(pairwise, loss, learner):
    import torch
    from neurec_amd.trainer import GeneralMFEngine
    from oracle import train
    rng = np.random.RandomState(abs(hash((pairwise, loss, learner))) % 1000)
    U, I, d, B, reg, lr = 200, 150, 64, 128, 0.01, 0.01
    P0 = (rng.randn(U, d) * 0.1).astype(np.float32)
    Q0 = (rng.randn(I, d) * 0.1).astype(np.float32)
    eng = GeneralMFEngine(P0, Q0, lr, reg, B, loss=loss, pairwise=pairwise, learner=learner)
    P, Q = P0.copy(), Q0.copy()
    P64, Q64 = P0.astype(np.float64), Q0.astype(np.float64)
    if learner == "adam":
        ad = train.Adam(lr)
        mP, vP, mQ, vQ = (np.zeros_like(x) for x in (P, P, Q, Q))
    else:
        oP, oQ = train.RowOptimizer(learner, lr, P.shape), train.RowOptimizer(learner, lr, Q.shape)
    loss2 = torch.zeros(2, device="cuda")
    for step in range(5):
        users = rng.randint(0, U, B).astype(np.int32)
        items = rng.randint(0, I, B).astype(np.int32)
        third = rng.randint(0, I, B).astype(np.int32) if pairwise else \
            (rng.rand(B) < 0.4).astype(np.float32)
        eng.step(_dev(users), _dev(items), _dev(third), loss2)
        l, r, dP, dQ = train.mf_general_loss_and_grads(P, Q, users, items, third, reg, pairwise, loss)
        l64, r64, _, _ = train.mf_general_loss_and_grads(P64, Q64, users, items,
                                                         third.astype(np.float64) if not pairwise else third,
                                                         reg, pairwise, loss)
        got = loss2.cpu().numpy()
        assert abs(got[0] - l) <= 2e-5 * max(abs(l), 1.0), (step, got, l, l64)
        assert abs(got[1] - r) <= 2e-5 * max(abs(r), 1e-3)
        if learner == "adam":
            ad.sparse_swept(P, mP, vP, dP); ad.sparse_swept(Q, mQ, vQ, dQ); ad.advance()
        else:
            oP.apply(P, dP, users)
            oQ.apply(Q, dQ, np.concatenate([items, third]) if pairwise else items)
        P64[:], Q64[:] = P, Q                                   # the twin follows the fp32 trajectory
    assert np.abs(eng.P.cpu().numpy() - P).max() < 2e-5
    assert np.abs(eng.Q.cpu().numpy() - Q).max() < 2e-5
    assert not eng.GP.cpu().numpy().any() and not eng.GQ.cpu().numpy().any()       # re-armed
    assert not eng.flagP.cpu().numpy().any() and not eng.flagQ.cpu().numpy().any()


def test_mf_variant_errors():
    from neurec_amd.trainer import GeneralMFEngine
    P, Q = np.zeros((4, 8), np.float32), np.zeros((5, 8), np.float32)
    with pytest.raises(Exception, match="suitable loss"):
        GeneralMFEngine(P, Q, 0.1, 0.0, 8, loss="cross_entropy", pairwise=True)
    with pytest.raises(ValueError, match="suitable optimizer"):
        GeneralMFEngine(P, Q, 0.1, 0.0, 8, learner="lbfgs")
