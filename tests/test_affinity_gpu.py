"""spmm_affinity_kernel (the cache-blocked full pass without phase barriers, csrc/spmm_blocked.hip) is
off by default (slower than the base kernel on MI355X, profiles/r02_exp_affinity.txt) but stays
correct: with NEUREC_SPMM_AFFINITY=1 the full pass and the fused-Adam hop give the base kernel's
results — bit-identical on rows of <= 64 non-zeros (strict ascending-column order in both), within
fp32 re-association on hub rows."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("window", [10 ** 9, 200000, 65536])
def test_affinity_schedule_equals_base_kernel(window, monkeypatch):
    import torch
    from neurec_amd import engine as E, synth
    from neurec_amd.graph import lightgcn_adjacency
    from oracle import train
    tr, _ = synth.interactions("gowalla", seed=5, scale=0.06)
    U, I = tr.shape
    coo = tr.tocoo()
    A = lightgcn_adjacency(coo.row, coo.col, U, I, "pre")
    rng = np.random.RandomState(0)
    X = rng.randn(U + I, 64).astype(np.float32)
    add, acc = rng.randn(U + I, 64).astype(np.float32), rng.randn(U + I, 64).astype(np.float32)
    base = E.SpmmCSR.from_scipy(A, split_row=U)
    base.ensure_schedule(64)
    assert E._lib.lib.nrhip_spmm_blocked_affinity(base.blocked, None) == 0
    monkeypatch.setenv("NEUREC_SPMM_AFFINITY", "1")
    monkeypatch.setenv("NEUREC_SPMM_AFF_BLOCK", str(window))
    aff = E.SpmmCSR.from_scipy(A, split_row=U)
    aff.ensure_schedule(64)
    wb = E.C.c_int(0)
    wa = E._lib.lib.nrhip_spmm_blocked_affinity(aff.blocked, E.C.byref(wb))
    assert wa >= 1 and (window > 10 ** 8 or wa > 1)
    outs = []
    for csr in (base, aff):
        Y, S = torch.empty(U + I, 64, device="cuda"), torch.empty(U + I, 64, device="cuda")
        csr.matmul(_dev(X), out=Y, addend=_dev(add), sum_in=_dev(acc), sum_out=S)
        outs.append((Y.cpu().numpy(), S.cpu().numpy()))
    short = np.diff(A.indptr) <= 64
    assert (~short).sum() > 0
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a[short], b[short])
        assert np.abs(a - b).max() < 1e-5
    want = train.spmm_rowwise(A, X) + add
    np.testing.assert_array_equal(outs[1][0][short], want[short])
    # fused-Adam hop on the affinity schedule == base schedule
    res = []
    for csr in (base, aff):
        var, m, v = _dev(acc.copy()), torch.zeros(U + I, 64, device="cuda"), torch.zeros(U + I, 64, device="cuda")
        st = E.AdamState(0.01)
        gb = _dev(np.zeros_like(add))
        E.call("nrhip_spmm_csr_adam", csr.plan, E._ptr(csr.indices), E._ptr(csr.vals), E._ptr(_dev(X)), 64,
               E._ptr(_dev(add)), E._ptr(gb), E._ptr(var), E._ptr(m), E._ptr(v), float(st.alpha()),
               float(st.beta1), float(st.beta2), float(st.eps), 0, E.C.c_void_p(0), E._stream())
        res.append(var.cpu().numpy())
    np.testing.assert_array_equal(res[0][short], res[1][short])
    assert np.abs(res[0] - res[1]).max() < 1e-5
