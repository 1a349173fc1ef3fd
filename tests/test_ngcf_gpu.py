"""NGCF layer kernels (fused dense half + SpMM + MFMA weight gradients) against oracle.train:
forward concat output, loss, and all parameters after several Adam steps, with the dropout
masks supplied as inputs (TF's Philox draw is not reproducible; the mask is data)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _problem(seed, U=300, I=260, d=16, L=2):
    from oracle import train
    rng = np.random.RandomState(seed)
    R = sp.random(U, I, 0.05, random_state=seed, format="csr", dtype=np.float32)
    R.data[:] = rng.randint(1, 6, R.nnz)                        # rating values, as ml-100k
    for h in range(2):                                           # two hub items (split rows at d=16)
        R = R.tolil(); R[rng.choice(U, int(U * 0.95), replace=False), h] = 3; R = R.tocsr()
    A = train.ngcf_adjacency(R, "norm")
    At = A.T.tocsr(); At.sort_indices()
    lim = np.sqrt(6.0 / (U + d))
    E0 = rng.uniform(-lim, lim, (U + I, d)).astype(np.float32)
    W = [((rng.randn(d, d) * 0.2).astype(np.float32), (rng.randn(1, d) * 0.05).astype(np.float32),
          (rng.randn(d, d) * 0.2).astype(np.float32), (rng.randn(1, d) * 0.05).astype(np.float32))
         for _ in range(L)]
    return rng, R, A, At, E0, W, U, I


def test_ngcf_forward_matches_oracle():
    from neurec_amd.trainer import NGCFEngine
    from oracle import train
    rng, R, A, At, E0, W, U, I = _problem(1)
    eng = NGCFEngine(A, At, U, I, E0, W, 0.001, 0.0, 0.1, 256)
    masks = [(rng.rand(U + I, 16) < 0.9).astype(np.uint8) for _ in W]
    out = eng.forward([_dev(m) for m in masks]).cpu().numpy()
    want, _ = train.ngcf_forward(A.astype(np.float64), E0.astype(np.float64),
                                 [tuple(w.astype(np.float64) for w in ws) for ws in W],
                                 [m.astype(np.float64) for m in masks], 0.9)
    assert np.abs(out - want).max() < 2e-6
    # device-drawn masks: right keep rate, rows normalised, reproducible per (seed, step)
    o1 = eng.forward().cpu().numpy()
    kept = np.mean([m.float().mean().item() for m in eng.mask])
    assert abs(kept - 0.9) < 0.01
    norms = np.linalg.norm(o1[:, 16:32], axis=1)
    assert np.all((np.abs(norms - 1) < 1e-5) | (norms == 0))


@pytest.mark.parametrize("reg,drop", [(0.0, 0.1), (0.01, 0.0)])
def test_ngcf_steps_track_oracle(reg, drop):
    import torch
    from neurec_amd.trainer import NGCFEngine
    from oracle import train
    rng, R, A, At, E0, W, U, I = _problem(2)
    B, lr, keep = 128, 0.005, 1.0 - drop
    eng = NGCFEngine(A, At, U, I, E0, W, lr, reg, drop, B)
    A64, At64 = A.astype(np.float64), At.astype(np.float64)
    oE = E0.astype(np.float64)
    oW = [[w.astype(np.float64) for w in ws] for ws in W]
    params = [oE] + [w for ws in oW for w in ws]
    ms, vs = [np.zeros_like(p) for p in params], [np.zeros_like(p) for p in params]
    ad = train.Adam(lr, dtype=np.float64)
    loss2 = torch.zeros(2, device="cuda")
    for step in range(5):
        bu, bp, bn = (rng.randint(0, n, B).astype(np.int32) for n in (U, I, I))
        masks = [(rng.rand(U + I, 16) < keep).astype(np.uint8) for _ in W]
        eng.step(_dev(bu), _dev(bp), _dev(bn), loss2, masks=[_dev(m) for m in masks])
        loss, dE, wg = train.ngcf_loss_and_grads(A64, At64, oE, [tuple(ws) for ws in oW],
                                                 [m.astype(np.float64) for m in masks], keep, U,
                                                 bu, bp, bn, reg)
        grads = [dE] + [g for gs in wg for g in gs]
        for p, m, v, g in zip(params, ms, vs, grads):
            ad.dense(p, m, v, g.reshape(p.shape))
        ad.advance()
        got = float(loss2.sum().item())
        assert abs(got - loss) <= 2e-5 * abs(loss), (step, got, loss)
    assert np.abs(eng.E0.cpu().numpy() - oE).max() < 2e-5
    for k in range(len(W)):
        for j in range(4):
            assert np.abs(eng.W[k][j].cpu().numpy().reshape(oW[k][j].shape) - oW[k][j]).max() < 5e-5, (k, j)
    assert not eng.dOut.cpu().numpy().any() and not eng.flag.cpu().numpy().any()
