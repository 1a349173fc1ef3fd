"""Mult-VAE at ANY p_dim on the GPU (neurec_amd/vae_wide.py; csrc/vae_wide.hip + csrc/gemm.hip) against
  * the reference's own MultiVAE class run under oracle/tf_shim.py at two, one and three layers
    (tests/golden/tfgraph_multivae_wide_*.npz; conf/MultiVAE.properties:3 lists [200, 600] and [200]), 1e-5;
  * oracle.train.multivae_general (pinned to the same fixtures on the CPU) at p_dim = [200, 600];
and the general fp32-MFMA GEMM against its own definition (the k-ascending fmaf chain) bit for bit."""
import json

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _err(got, want):
    return float(np.abs(np.asarray(got, np.float64) - want).max())


def _rel(got, want):
    return float(np.max(np.abs(np.asarray(got, np.float64) - want) / np.abs(want)))


def _drop_by_entry(R, rows, mask):
    out = np.ones(R.nnz, np.float32)
    for b, u in enumerate(rows):
        lo, hi = R.indptr[u], R.indptr[u + 1]
        out[lo:hi] = mask[b, R.indices[lo:hi]]
    return out


def _names(n):
    return ["Wq%d" % i for i in range(n)] + ["bq%d" % i for i in range(n)] + \
           ["Wp%d" % i for i in range(n)] + ["bp%d" % i for i in range(n)]


@pytest.mark.parametrize("tag", ["24x40", "20", "8x16x24"])
def test_wide_engine_equals_the_reference_graph(tag):
    from neurec_amd import engine as E
    from neurec_amd.vae_wide import MultiVAEWideEngine
    g = load_golden("tfgraph_multivae_wide_" + tag)
    h = json.loads(str(g["hyper"]))
    U, I, n = int(g["n_users"]), int(g["n_items"]), int(g["n_layers"])
    R = sp.csr_matrix((np.ones(len(g["train_indices"]), np.float32), g["train_indices"], g["train_indptr"]),
                      shape=(U, I))
    P = lambda k: [g["%s%d_0" % (k, i)] for i in range(n)]
    eng = MultiVAEWideEngine(E.DeviceCSR.from_scipy(R), I, P("Wq"), P("bq"), P("Wp"), P("bp"), h["learning_rate"],
                             h["reg"], h["activation"], h["batch_size"])
    got, first = [], None
    for s, rows in enumerate(g["rows"]):
        if first is None:                                    # gradients of the first step, before Adam consumes them
            eng.step(_dev(rows.astype(np.int32)), float(g["anneal"][s]), 0.8,
                     drop_given=_dev(_drop_by_entry(R, rows, g["drop_masks"][s])),
                     eps_given=_dev(g["eps"][s].astype(np.float32)), apply=False)
            first = [x.cpu().numpy().copy() for x in eng.G]
            eng.G[0].zero_()
        eng.step(_dev(rows.astype(np.int32)), float(g["anneal"][s]), 0.8,
                 drop_given=_dev(_drop_by_entry(R, rows, g["drop_masks"][s])),
                 eps_given=_dev(g["eps"][s].astype(np.float32)))
        got.append(eng.loss()[0])
    assert _rel(got, g["f32_loss"]) <= TOL and _rel(got, g["f64_loss"]) <= TOL
    names = _names(n)
    dgrad = max(_err(a.reshape(g["f64_d" + k].shape), g["f64_d" + k]) for a, k in zip(first, names))
    bar = max(_err(g["f32_" + k], g["f64_" + k]) for k in names)
    d = max(_err(p.cpu().numpy().reshape(g["f64_" + k].shape), g["f64_" + k]) for p, k in zip(eng.params, names))
    # predict(): the reference's accumulating row (MultiVAE.py:195-203) as one extra CSR
    acc, rows_acc = set(), []
    for u in g["ratings_users"]:
        acc |= set(R[u].indices.tolist())
        rows_acc.append(sorted(acc))
    indptr = np.cumsum([0] + [len(r) for r in rows_acc])
    Racc = sp.csr_matrix((np.ones(indptr[-1], np.float32), np.concatenate(rows_acc), indptr), shape=(len(rows_acc), I))
    out = eng.logits(_dev(np.arange(len(rows_acc), dtype=np.int32)), E.DeviceCSR.from_scipy(Racc))
    dr = _err(out[:, :I].cpu().numpy(), g["f64_ratings"])
    print("wide Mult-VAE %s: first-step gradients %.1e, parameters after %d steps %.1e (reference fp32-vs-fp64 %.1e), "
          "predict() %.1e" % (tag, dgrad, len(got), d, bar, dr))
    assert dgrad <= TOL and d <= TOL + bar and dr <= TOL + bar


def test_wide_engine_at_p_dim_200_600_equals_the_restatement():
    """conf/MultiVAE.properties:3's own alternative: p_dim = [200, 600] (q: I -> 600 -> 400, p: 200 -> 600 -> I)."""
    from neurec_amd import engine as E
    from neurec_amd.vae_wide import MultiVAEWideEngine
    from oracle import train as O
    rng = np.random.RandomState(5)
    U, I, B, z, hdim = 300, 3000, 96, 200, 600
    R = sp.random(U, I, density=0.02, random_state=rng, format="csr", dtype=np.float32)
    R.data[:] = 1
    R = R[np.diff(R.indptr) > 0]
    U = R.shape[0]
    xav = lambda a, b: (rng.uniform(-1, 1, (a, b)) * np.sqrt(6.0 / (a + b))).astype(np.float32)
    tn = lambda b: (rng.randn(b) * 0.001).astype(np.float32)
    Wq, bq = [xav(I, hdim), xav(hdim, 2 * z)], [tn(hdim), tn(2 * z)]
    Wp, bp = [xav(z, hdim), xav(hdim, I)], [tn(hdim), tn(I)]
    lr, reg = 1e-3, 0.01
    eng = MultiVAEWideEngine(E.DeviceCSR.from_scipy(R), I, Wq, bq, Wp, bp, lr, reg, "tanh", B)
    for dt, tol in ((np.float64, TOL),):
        W = [[x.astype(dt) for x in grp] for grp in (Wq, bq, Wp, bp)]
        params = W[0] + W[1] + W[2] + W[3]
        ms, vs = [np.zeros_like(x) for x in params], [np.zeros_like(x) for x in params]
        adam = O.Adam(lr, dtype=dt)
        want, got = [], []
        for s in range(3):
            rows = rng.choice(U, B, replace=False).astype(np.int32)
            mask = (rng.rand(B, I) < 0.8).astype(np.float32)
            eps = (rng.randn(B, z) * 0.01).astype(np.float32)
            X = np.asarray(R[rows].todense(), dtype=dt)
            loss, grads, _, _ = O.multivae_general(X, W[0], W[1], W[2], W[3], mask.astype(dt), dt(0.8), eps.astype(dt),
                                                   0.2 * (s + 1), reg, "tanh")
            if s == 0:
                eng.step(_dev(rows), 0.2, 0.8, drop_given=_dev(_drop_by_entry(R, rows, mask)), eps_given=_dev(eps),
                         apply=False)
                flat = grads[0] + grads[1] + grads[2] + grads[3]
                dgrad = max(_err(a.cpu().numpy().reshape(b.shape), b) / max(np.abs(b).max(), 1e-30)
                            for a, b in zip(eng.G, flat))
                eng.G[0].zero_()
            for pp, m, v, gg in zip(params, ms, vs, grads[0] + grads[1] + grads[2] + grads[3]):
                adam.dense(pp, m, v, gg.reshape(pp.shape))
            adam.advance()
            want.append(float(loss))
            eng.step(_dev(rows), 0.2 * (s + 1), 0.8, drop_given=_dev(_drop_by_entry(R, rows, mask)), eps_given=_dev(eps))
            got.append(eng.loss()[0])
        d = max(_err(p.cpu().numpy().reshape(w.shape), w) for p, w in zip(eng.params, params))
        print("wide Mult-VAE [200, 600]: loss %.1e, first-step gradients (relative to each one's max) %.1e, "
              "parameters after 3 steps %.1e" % (_rel(got, want), dgrad, d))
        assert _rel(got, want) <= tol and dgrad <= tol
        # Adam's first steps move every coordinate by ~lr whatever |g|: coordinates whose gradient fp32 does not
        # resolve may land on the other side ("Adam and tiny gradients"); the bulk must agree
        assert d <= 2.5 * lr


def test_wide_engine_steps_a_batch_beyond_2048_rows():
    """the column sums of a batch of more than 2,048 rows go through a workspace (nrhip_colsum_rows'
    chunked form) — the engine owns one; first-step gradients against the restatement at B = 2,304."""
    from neurec_amd import engine as E
    from neurec_amd.vae_wide import MultiVAEWideEngine
    from oracle import train as O
    rng = np.random.RandomState(9)
    U, I, B, z, hdim = 2400, 500, 2304, 12, 40
    R = sp.random(U, I, density=0.03, random_state=rng, format="csr", dtype=np.float32)
    R.data[:] = 1
    R = R[np.diff(R.indptr) > 0]
    U = R.shape[0]
    B = min(B, U)
    assert B > 2048
    xav = lambda a, b: (rng.uniform(-1, 1, (a, b)) * np.sqrt(6.0 / (a + b))).astype(np.float32)
    tn = lambda b: (rng.randn(b) * 0.001).astype(np.float32)
    Wq, bq = [xav(I, hdim), xav(hdim, 2 * z)], [tn(hdim), tn(2 * z)]
    Wp, bp = [xav(z, hdim), xav(hdim, I)], [tn(hdim), tn(I)]
    eng = MultiVAEWideEngine(E.DeviceCSR.from_scipy(R), I, Wq, bq, Wp, bp, 1e-3, 0.0, "tanh", B)
    rows = rng.choice(U, B, replace=False).astype(np.int32)
    mask = (rng.rand(B, I) < 0.8).astype(np.float32)
    eps = (rng.randn(B, z) * 0.01).astype(np.float32)
    dt = np.float64
    W = [[x.astype(dt) for x in grp] for grp in (Wq, bq, Wp, bp)]
    X = np.asarray(R[rows].todense(), dtype=dt)
    loss, grads, _, _ = O.multivae_general(X, W[0], W[1], W[2], W[3], mask.astype(dt), dt(0.8), eps.astype(dt), 0.2, 0.0,
                                           "tanh")
    eng.step(_dev(rows), 0.2, 0.8, drop_given=_dev(_drop_by_entry(R, rows, mask)), eps_given=_dev(eps), apply=False)
    flat = grads[0] + grads[1] + grads[2] + grads[3]
    dgrad = max(_err(a.cpu().numpy().reshape(b.shape), b) / max(np.abs(b).max(), 1e-30) for a, b in zip(eng.G, flat))
    assert abs(eng.loss()[0] - float(loss)) <= TOL * abs(float(loss)) and dgrad <= TOL, (eng.loss()[0], float(loss), dgrad)


@pytest.mark.parametrize("M,N,K,splits", [(64, 64, 16, 1), (100, 333, 77, 1), (512, 1000, 600, 1), (200, 600, 512, 1),
                                          (96, 40, 5000, 8), (512, 600, 4099, 16), (1, 1, 1, 1)])
def test_gemm_kmajor_is_the_k_ascending_fmaf_chain(M, N, K, splits):
    import ctypes as C
    import torch
    from neurec_amd._lib import call
    from neurec_amd.engine import _ptr, _stream
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(K, M).astype(np.float32)
    Bm = rng.randn(K, N).astype(np.float32)
    dA, dB = _dev(A), _dev(Bm)
    out = torch.full((M, N + 3), 7.0, dtype=torch.float32, device="cuda")
    nbytes = C.c_size_t(0)
    call("nrhip_gemm_workspace_bytes", M, N, splits, C.byref(nbytes))
    ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device="cuda")
    call("nrhip_gemm_kmajor", _ptr(dA), M, _ptr(dB), N, M, N, K, _ptr(out), N + 3, 0, None, -1, splits, _ptr(ws),
         ws.numel() if splits > 1 else 0, _stream())
    got = out.cpu().numpy()
    assert (got[:, N:] == 7.0).all()
    want64 = A.T.astype(np.float64) @ Bm.astype(np.float64)
    if splits == 1:
        from oracle import native
        want = native.score_gemm(np.ascontiguousarray(A.T), None, np.ascontiguousarray(Bm.T))   # the true fmaf chain
        assert np.array_equal(got[:, :N], want)
    assert np.abs(got[:, :N] - want64).max() <= 2e-6 * np.sqrt(K) * max(1.0, np.abs(want64).max())
    # accumulate: C += A^T B continues the chain from C
    call("nrhip_gemm_kmajor", _ptr(dA), M, _ptr(dB), N, M, N, K, _ptr(out), N + 3, 1, None, -1, splits, _ptr(ws),
         ws.numel() if splits > 1 else 0, _stream())
    assert np.abs(out.cpu().numpy()[:, :N] - 2 * want64).max() <= 4e-6 * np.sqrt(K) * max(1.0, np.abs(want64).max())
    # transpose
    T = torch.zeros((N, K + 1), dtype=torch.float32, device="cuda")
    call("nrhip_transpose2d", _ptr(dB), N, K, N, _ptr(T), K + 1, _stream())
    assert np.array_equal(T.cpu().numpy()[:, :K], Bm.T)


@pytest.mark.parametrize("M,N,K,splits,act", [(96, 130, 50, 1, 2), (300, 700, 129, 4, 0), (512, 40981, 40, 1, -1)])
def test_gemm_epilogue_is_bias_then_activation(M, N, K, splits, act):
    """a dense layer y = act(x W + b) as one call (MultiVAE.py:79-84,124-129: tf.matmul + bias, then the activation)"""
    import ctypes as C
    import torch
    from neurec_amd._lib import call
    from neurec_amd.engine import _ptr, _stream
    from oracle import native
    rng = np.random.RandomState(M + N + K)
    A, Bm, bias = rng.randn(K, M).astype(np.float32), rng.randn(K, N).astype(np.float32), rng.randn(N).astype(np.float32)
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    nbytes = C.c_size_t(0)
    call("nrhip_gemm_workspace_bytes", M, N, splits, C.byref(nbytes))
    ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device="cuda")
    dA, dB, dbias = _dev(A), _dev(Bm), _dev(bias)
    call("nrhip_gemm_kmajor", _ptr(dA), M, _ptr(dB), N, M, N, K, _ptr(out), N, 0, _ptr(dbias), act,
         splits, _ptr(ws), ws.numel() if splits > 1 else 0, _stream())
    pre = native.score_gemm(np.ascontiguousarray(A.T), None, np.ascontiguousarray(Bm.T)) + bias[None, :]
    want = {2: np.maximum(pre, 0), 0: np.tanh(pre.astype(np.float64)), -1: pre}[act]
    got = out.cpu().numpy()
    if splits == 1 and act != 0:
        assert np.array_equal(got, want)                     # the fmaf chain, one rounded add of the bias, max(., 0)
    else:
        assert np.abs(got - want).max() <= 2e-6 * np.sqrt(K)


def test_gemm_edge_cases():
    """empty contraction (C = bias, activation applied), empty outputs, accumulate over an empty contraction, and the
    leading-dimension bound of the buffer-addressed operand loads"""
    import torch
    from neurec_amd._lib import call
    from neurec_amd.engine import _ptr, _stream
    A = torch.zeros((1, 8), device="cuda")
    Bm = torch.zeros((1, 5), device="cuda")
    bias = torch.tensor([-2.0, -1.0, 0.0, 1.0, 2.0], device="cuda")
    out = torch.full((8, 5), 3.0, device="cuda")
    call("nrhip_gemm_kmajor", _ptr(A), 8, _ptr(Bm), 5, 8, 5, 0, _ptr(out), 5, 0, _ptr(bias), 2, 1, None, 0, _stream())
    assert torch.equal(out, torch.tensor([0.0, 0.0, 0.0, 1.0, 2.0], device="cuda").expand(8, 5))
    out.fill_(3.0)
    call("nrhip_gemm_kmajor", _ptr(A), 8, _ptr(Bm), 5, 8, 5, 0, _ptr(out), 5, 1, None, -1, 1, None, 0, _stream())
    assert torch.equal(out, torch.full((8, 5), 3.0, device="cuda"))
    call("nrhip_gemm_kmajor", _ptr(A), 8, _ptr(Bm), 5, 0, 5, 1, _ptr(out), 5, 0, None, -1, 1, None, 0, _stream())   # M = 0
    with pytest.raises(ValueError):
        call("nrhip_gemm_kmajor", _ptr(A), 1 << 24, _ptr(Bm), 5, 8, 5, 1, _ptr(out), 5, 0, None, -1, 1, None, 0, _stream())
    with pytest.raises(NotImplementedError, match="k-minor operand"):        # 2.5 GB through one 2 GB buffer window
        call("nrhip_gemm_f32", _ptr(A), 600, 1, _ptr(Bm), 5, 0, 1 << 20, 5, 600, _ptr(out), 5, 0, None, -1, 1, None, 0,
             _stream())
    with pytest.raises(Exception):                           # splits without a workspace
        call("nrhip_gemm_kmajor", _ptr(A), 8, _ptr(Bm), 5, 8, 5, 1, _ptr(out), 5, 0, None, -1, 4, None, 0, _stream())


@pytest.mark.parametrize("M,N,K,splits", [(100, 333, 77, 1), (512, 700, 600, 1), (130, 64, 1000, 4), (64, 64, 5000, 16),
                                          (1, 1, 1, 1), (257, 129, 17, 1)])
@pytest.mark.parametrize("a_kminor,b_kminor", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_operand_layouts_give_the_same_chain(M, N, K, splits, a_kminor, b_kminor):
    """nrhip_gemm_f32: either operand k-major ([K][ld]) or k-minor ([M][ld] / [N][ld], contraction index contiguous) —
    x W, x W^T, x^T g without transposed copies; every layout is the same k-ascending fmaf chain per element."""
    import ctypes as C
    import torch
    from neurec_amd._lib import call
    from neurec_amd.engine import _ptr, _stream
    from oracle import native
    rng = np.random.RandomState(M + N + K + 2 * a_kminor + b_kminor)
    A = rng.randn(M, K).astype(np.float32)                  # a(k, m) = A[m][k]
    Bm = rng.randn(N, K).astype(np.float32)                 # b(k, n) = Bm[n][k]
    pad = 3
    hA = np.zeros((M, K + pad), np.float32) if a_kminor else np.zeros((K, M + pad), np.float32)
    hB = np.zeros((N, K + pad), np.float32) if b_kminor else np.zeros((K, N + pad), np.float32)
    if a_kminor: hA[:, :K] = A
    else: hA[:, :M] = A.T
    if b_kminor: hB[:, :K] = Bm
    else: hB[:, :N] = Bm.T
    dA, dB = _dev(hA), _dev(hB)
    out = torch.full((M, N + 2), 5.0, dtype=torch.float32, device="cuda")
    nbytes = C.c_size_t(0)
    call("nrhip_gemm_workspace_bytes", M, N, splits, C.byref(nbytes))
    ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device="cuda")
    call("nrhip_gemm_f32", _ptr(dA), hA.shape[1], a_kminor, _ptr(dB), hB.shape[1], b_kminor, M, N, K, _ptr(out), N + 2, 0,
         None, -1, splits, _ptr(ws), ws.numel() if splits > 1 else 0, _stream())
    got = out.cpu().numpy()
    assert (got[:, N:] == 5.0).all()
    if splits == 1:
        np.testing.assert_array_equal(got[:, :N], native.score_gemm(A, None, Bm))
    else:
        want = A.astype(np.float64) @ Bm.astype(np.float64).T
        assert np.abs(got[:, :N] - want).max() <= 2e-6 * np.sqrt(K) * max(1.0, np.abs(want).max())
