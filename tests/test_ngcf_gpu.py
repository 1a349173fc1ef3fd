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
    """Five Adam steps on E0 and all eight weight tensors.  The pin is the fp32 restatement
    (north_star's 1e-5); the fp64 twin shows how far fp32 arithmetic itself drifts on this
    problem (Adam moves every coordinate by ~lr·sign(g): coordinates whose gradient is at
    rounding level go either way), and only that measured distance is granted on top."""
    import torch
    from neurec_amd.trainer import NGCFEngine
    from oracle import train
    rng, R, A, At, E0, W, U, I = _problem(2)
    B, lr, keep = 128, 0.005, 1.0 - drop
    eng = NGCFEngine(A, At, U, I, E0, W, lr, reg, drop, B)
    steps = [(tuple(rng.randint(0, n, B).astype(np.int32) for n in (U, I, I)),
              [(rng.rand(U + I, 16) < keep).astype(np.uint8) for _ in W]) for _ in range(5)]
    loss2 = torch.zeros(2, device="cuda")
    got_loss = []
    for (bu, bp, bn), masks in steps:
        eng.step(_dev(bu), _dev(bp), _dev(bn), loss2, masks=[_dev(m) for m in masks])
        got_loss.append(float(loss2.sum().item()))

    def run(dt):
        A_, At_ = A.astype(dt), At.astype(dt)
        oE = E0.astype(dt)
        oW = [[w.astype(dt) for w in ws] for ws in W]
        params = [oE] + [w for ws in oW for w in ws]
        ms, vs = [np.zeros_like(p) for p in params], [np.zeros_like(p) for p in params]
        ad = train.Adam(lr, dtype=dt)
        losses = []
        for (bu, bp, bn), masks in steps:
            loss, dE, wg = train.ngcf_loss_and_grads(A_, At_, oE, [tuple(ws) for ws in oW],
                                                     [m.astype(dt) for m in masks], keep, U, bu, bp, bn, reg)
            for p, m, v, g in zip(params, ms, vs, [dE] + [g for gs in wg for g in gs]):
                ad.dense(p, m, v, g.reshape(p.shape))
            ad.advance()
            losses.append(float(loss))
        return np.asarray(losses), params
    l32, p32 = run(np.float32)
    l64, p64 = run(np.float64)
    got = [eng.E0.cpu().numpy()] + [eng.W[k][j].cpu().numpy() for k in range(len(W)) for j in range(4)]
    d32 = max(np.abs(g.reshape(a.shape) - a).max() for g, a in zip(got, p32))
    d64 = max(np.abs(g.reshape(a.shape) - a).max() for g, a in zip(got, p64))
    bar = max(np.abs(a.astype(np.float64) - b).max() for a, b in zip(p32, p64))
    dl32 = (np.abs(np.asarray(got_loss) - l32) / np.abs(l32)).max()
    dl64 = (np.abs(np.asarray(got_loss) - l64) / np.abs(l64)).max()
    barl = (np.abs(l32 - l64) / np.abs(l64)).max()
    print("NGCF reg=%g drop=%g: loss rel err vs fp32 oracle %.1e, vs fp64 %.1e (oracle fp32-vs-fp64 %.1e); "
          "parameters max abs err vs fp32 oracle %.1e, vs fp64 %.1e (oracle fp32-vs-fp64 %.1e)"
          % (reg, drop, dl32, dl64, barl, d32, d64, bar))
    assert dl32 <= 1e-5 and dl64 <= 1e-5 + barl
    assert d32 <= 1e-5 + bar and d64 <= 1e-5 + bar        # fp32 runs differ from each other by <= bar too
    assert not eng.dOut.cpu().numpy().any() and not eng.flag.cpu().numpy().any()


def test_native_step_equals_the_spelled_out_launch_sequence():
    """nrhip_ngcf_step (one call) against NGCFEngine's Python launch sequence (what runs with more than
    NGCF_MAX_LAYERS layers): given masks and device-drawn ones, three steps, every trainable and both
    Adam moments bit for bit."""
    import torch
    from neurec_amd import engine as E
    from neurec_amd.trainer import NGCFEngine
    rng, R, A, At, E0, W, U, I = _problem(7)
    native = NGCFEngine(A, At, U, I, E0, W, 0.003, 0.01, 0.1, 256)
    spelled = NGCFEngine(A, At, U, I, E0, W, 0.003, 0.01, 0.1, 256)
    assert native._ctx is not None
    spelled._ctx = None
    la, lb = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    for step in range(3):
        B = 200
        bu, bp, bn = (_dev(rng.randint(0, n, B).astype(np.int32)) for n in (U, I, I))
        masks = None if step == 1 else [_dev((rng.rand(U + I, 16) < 0.9).astype(np.uint8)) for _ in W]
        plan = E.bpr_plan(bu, bp, bn, B, U) if step == 2 else None
        native.step(bu, bp, bn, la, masks=masks, plan=plan)
        spelled.step(bu, bp, bn, lb, masks=masks, plan=plan)
        assert torch.equal(la, lb)
    assert torch.equal(native.E0, spelled.E0) and torch.equal(native.mE, spelled.mE) and torch.equal(native.vE, spelled.vE)
    for k in range(2):
        for j in range(4):
            assert torch.equal(native.W[k][j], spelled.W[k][j]), (k, j)
            assert torch.equal(native.vW[k][j], spelled.vW[k][j]), (k, j)
    assert torch.equal(native.forward(), spelled.forward())          # same (seed, step) -> same draw
