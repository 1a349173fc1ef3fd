"""Mult-VAE for ANY p_dim on the GPU (model/general_recommender/MultiVAE.py:46-135; conf/MultiVAE.properties:3 lists
[200, 600] and [200] next to the shipped [16, 32], and the class builds len(p_dim) layers each way).

trainer.MultiVAEEngine is the register-resident form for the shipped two-layer shape with widths <= 32; this engine
strings width-generic kernels (csrc/vae_wide.hip) and a general fp32-MFMA GEMM (csrc/gemm.hip) into the same step:

  forward   first encoder layer as a bag-sum over the user's CSR row (no dense [B][I] input) -> dense layers ->
            [mu | logvar] -> z = mu + eps * exp(logvar / 2) -> dense layers -> logits = g W_last + b on the matrix cores
            (g row-major and the TF variable [h][I], each as it lies)
  backward  dLoss/dlogits in place on the logits slab; dW_last = g^T D; d g = D W_last^T (split over the 40,981-long
            contraction) — every operand read in the layout it is stored in (nrhip_gemm_f32's k-major / k-minor
            operand forms: no transposed copies); dense layers back; the first encoder layer's gradient scattered
            along the CSR rows
  update    TF's dense ApplyAdam on every variable (MultiVAE.py:137-139)

Weights are kept in TensorFlow's layout ([in][out]).  Checked against the reference class itself at one, two and three
layers (tests/golden/tfgraph_multivae_wide_*.npz) and against oracle.train.multivae_general at [200, 600].
"""
import numpy as np
import torch

from . import engine as E
from ._lib import call
from .engine import _ptr, _stream


def _act_id(act, last):
    return -1 if last else E.VAE_ACTS[act]


class MultiVAEWideEngine:
    def __init__(self, train_csr, n_items, Wq, bq, Wp, bp, lr, reg, act, max_batch, seed=2017, learner="adam"):
        dev = E.require_gpu()
        self.csr, self.n_items = train_csr, int(n_items)
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)
        self.n = len(Wq)
        assert len(bq) == len(Wp) == len(bp) == self.n >= 1
        self.Wq, self.bq = [f(w) for w in Wq], [f(np.reshape(b, -1)) for b in bq]
        self.Wp, self.bp = [f(w) for w in Wp], [f(np.reshape(b, -1)) for b in bp]
        self.z = self.Wp[0].shape[0]
        assert self.Wq[0].shape[0] == self.n_items and self.Wp[-1].shape[1] == self.n_items
        assert self.Wq[-1].shape[1] == 2 * self.z
        if act not in E.VAE_ACTS:
            raise NotImplementedError("activation %r is not built (tanh/sigmoid/relu/identity)" % act)
        self.act, self.reg, self.seed, self.t = act, float(reg), int(seed), 0
        self.adam = E.AdamState(lr)
        self.B = int(max_batch)
        B, I = self.B, self.n_items
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.params = self.Wq + self.bq + self.Wp + self.bp
        self.G = [torch.zeros_like(p) for p in self.params]
        self.M = [torch.zeros_like(p) for p in self.params]
        self.V = [torch.zeros_like(p) for p in self.params]
        self.learner = E.make_learner(learner, lr)          # learner.py:2-17; None: adam
        if self.learner is not None:
            self.learner.init_slots(self.M, self.V)
        self.Hq = [z(B, w.shape[1]) for w in self.Wq]              # encoder layer outputs (the last: [mu | logvar])
        self.Gp = [z(B, w.shape[1]) for w in self.Wp[:-1]]         # decoder hidden outputs
        self.dHq = [z(B, w.shape[1]) for w in self.Wq]
        self.dGp = [z(B, w.shape[1]) for w in self.Wp[:-1]]
        self.EPSSTD, self.ZS, self.dZ = z(B, self.z), z(B, self.z), z(B, self.z)
        self.KLb, self.nll = z(B), z(B)
        self.h0val = z(max(train_csr.nnz, 1))
        self.h_last = self.Wp[-1].shape[0]                          # width feeding the item layer
        self.ld = (I + 63) // 64 * 64
        self.S = z(B, self.ld)                                      # logits slab, then dLoss/dlogits in place
        self.splits = 32                                            # of the 40,981-long contraction of d g
        mids = self.Wq[1:] + self.Wp[:-1]                            # the layers that are plain dense products
        wmax = max([self.h_last] + [max(w.shape) for w in mids])
        self.mid_splits = 4                                         # contraction cuts of the small products
        nbytes = E.C.c_size_t(0)
        call("nrhip_gemm_workspace_bytes", max(B, wmax), wmax, max(self.splits, self.mid_splits), E.C.byref(nbytes))
        self.ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=dev)
        # column sums over more than 2,048 rows go through ceil(rows / 512) partial rows (nrhip_colsum_rows)
        widths = [I, self.h_last] + [w.shape[1] for w in mids] + [self.Wq[0].shape[1]]
        self.cs_ws = z(max((B + 511) // 512 * max(widths), 64)) if B > 2048 else None
        self.stats = z(2)
        self.regsum = torch.zeros(1, dtype=torch.float64, device=dev)
        self.last_anneal = 0.0

    # ------------------------------------------------------------------ pieces
    def _gemm(self, A, lda, a_kminor, Bm, ldb, b_kminor, M, N, K, Cm, ldc, splits=1, bias=None, act=-1):
        """C[m][n] = sum_k a(k, m) b(k, n) (+ bias, activation); an operand is k-major ([K][ld]) or k-minor ([M|N][ld])"""
        call("nrhip_gemm_f32", _ptr(A), int(lda), int(a_kminor), _ptr(Bm), int(ldb), int(b_kminor), int(M), int(N),
             int(K), _ptr(Cm), int(ldc), 0, _ptr(bias, torch.float32, allow_none=True), int(act), int(splits),
             _ptr(self.ws), self.ws.numel() if splits > 1 else 0, _stream())

    def _colsum(self, X, ld, rows, cols, out):
        ws = self.cs_ws
        call("nrhip_colsum_rows", _ptr(X), int(ld), int(rows), int(cols), _ptr(out),
             _ptr(ws, allow_none=True), 0 if ws is None else ws.numel() * 4, _stream())

    def _dense_fwd(self, X, W, b, B, act, Y):
        """Y[:B] = act(X[:B] W + b): x row-major (k-minor) and the TF variable [in][out] (k-major), as they lie."""
        K, N = W.shape
        self._gemm(X, K, 1, W, N, 0, B, N, K, Y, N, bias=b, act=act)

    def _dense_bwd(self, dA, X, W, B, dX, dW, db):
        """dW = X^T dA (contraction over the batch rows: both k-major as stored), db = column sums, dX = dA W^T (both
        k-minor as stored)."""
        K, N = W.shape
        self._gemm(X, K, 0, dA, N, 0, K, N, B, dW, N, splits=self.mid_splits if B >= 256 else 1)
        self._colsum(dA, N, B, N, db)
        self._gemm(dA, N, 1, W, N, 1, B, K, N, dX, K)

    def _forward(self, rows, csr, keep, is_training, drop_given, eps_given, S):
        """fills Hq / ZS / Gp for the B = rows.numel() rows and the logits slab S[:B]"""
        B, n = rows.numel(), self.n
        call("nrhip_vae_bag_fwd", _ptr(csr.indptr), _ptr(csr.indices), _ptr(rows, torch.int32), B,
             self.Wq[0].shape[1], _ptr(self.Wq[0]), _ptr(self.bq[0]), _act_id(self.act, n == 1), float(keep),
             _ptr(drop_given, torch.float32, allow_none=True), self.seed, self.t,
             _ptr(self.h0val) if csr is self.csr else None, _ptr(self.Hq[0]), _stream())
        for i in range(1, n):
            self._dense_fwd(self.Hq[i - 1], self.Wq[i], self.bq[i], B, _act_id(self.act, i == n - 1), self.Hq[i])
        call("nrhip_vae_sample", _ptr(self.Hq[-1]), B, self.z, _ptr(eps_given, torch.float32, allow_none=True),
             float(is_training), self.seed, self.t, _ptr(self.EPSSTD), _ptr(self.ZS), _ptr(self.KLb), _stream())
        g = self.ZS
        for i in range(n - 1):
            self._dense_fwd(g, self.Wp[i], self.bp[i], B, E.VAE_ACTS[self.act], self.Gp[i])
            g = self.Gp[i]
        h, I = self.h_last, self.n_items
        self._gemm(g, h, 1, self.Wp[-1], I, 0, B, I, h, S, S.stride(0), bias=self.bp[-1])   # logits + bias
        return g

    def logits(self, rows, csr=None, out=None):
        """p-network output for the given CSR rows at is_training = 0, keep_prob = 1 (MultiVAE.py:186-206)."""
        csr = self.csr if csr is None else csr
        Bt = rows.numel()
        if out is None:
            out = torch.empty((Bt, self.ld), dtype=torch.float32, device=self.S.device)
        for lo in range(0, Bt, self.B):
            n = min(self.B, Bt - lo)
            self._forward(rows[lo:lo + n].contiguous(), csr, 1.0, 0.0, None, None, out[lo:lo + n])
        return out[:Bt]

    def gemm_out(self, rows):
        return torch.empty((rows, self.ld), dtype=torch.float32, device=self.S.device)

    # ------------------------------------------------------------------ one optimiser step
    def step(self, rows, anneal, keep=0.8, drop_given=None, eps_given=None, want_loss=True, apply=True):
        B, n, I, h = rows.numel(), self.n, self.n_items, self.h_last
        if B > self.B or B < 1:
            raise ValueError("batch size %d outside [1, %d]" % (B, self.B))
        csr = self.csr
        g_last = self._forward(rows, csr, keep, 1.0, drop_given, eps_given, self.S)
        S, ld = self.S, self.ld
        call("nrhip_vae_softmax_dlogits", _ptr(S), ld, B, I, _ptr(csr.indptr), _ptr(csr.indices),
             _ptr(rows, torch.int32), _ptr(self.nll), _stream())                           # S is D from here on
        iWq, ibq, iWp, ibp = 0, n, 2 * n, 3 * n                                           # offsets into params / G
        # last decoder layer on the matrix cores
        self._gemm(g_last, h, 0, S, ld, 0, h, I, B, self.G[iWp + n - 1], I)                # dW = g^T D
        self._colsum(S, ld, B, I, self.G[ibp + n - 1])
        dg = self.dGp[-1] if n > 1 else self.dZ
        self._gemm(S, ld, 1, self.Wp[-1], I, 1, B, h, I, dg, h, splits=self.splits)        # d g = D W^T
        # hidden decoder layers
        for i in range(n - 2, -1, -1):
            K, N = self.Wp[i].shape
            call("nrhip_act_bwd", _ptr(dg), _ptr(self.Gp[i]), B * N, E.VAE_ACTS[self.act], _ptr(dg), _stream())
            x = self.Gp[i - 1] if i > 0 else self.ZS
            dx = self.dGp[i - 1] if i > 0 else self.dZ
            self._dense_bwd(dg, x, self.Wp[i], B, dx, self.G[iWp + i], self.G[ibp + i])
            dg = dx
        call("nrhip_vae_sample_bwd", _ptr(self.dZ), _ptr(self.Hq[-1]), _ptr(self.EPSSTD), B, self.z, float(anneal),
             _ptr(self.dHq[-1]), _stream())
        d = self.dHq[-1]
        for i in range(n - 1, 0, -1):
            K, N = self.Wq[i].shape
            self._dense_bwd(d, self.Hq[i - 1], self.Wq[i], B, self.dHq[i - 1], self.G[iWq + i], self.G[ibq + i])
            d = self.dHq[i - 1]
            call("nrhip_act_bwd", _ptr(d), _ptr(self.Hq[i - 1]), B * K, E.VAE_ACTS[self.act], _ptr(d), _stream())
        w0 = self.Wq[0].shape[1]
        call("nrhip_vae_dwq0_wide", _ptr(csr.indptr), _ptr(csr.indices), _ptr(rows, torch.int32), B, w0,
             _ptr(self.h0val), _ptr(d), _ptr(self.G[iWq]), _stream())                     # G[Wq0] is zero here
        self._colsum(d, w0, B, w0, self.G[ibq])
        if want_loss:
            E.mean2_f32(self.nll[:B], self.KLb[:B], self.stats)
        if self.reg != 0.0:
            if want_loss:
                self.regsum.zero_()
            for k in list(range(iWq, iWq + n)) + list(range(iWp, iWp + n)):
                if want_loss:
                    E.sumsq_accumulate(self.params[k], self.regsum)
                E.axpy(2.0 * self.reg, self.params[k], self.G[k])
        self.last_anneal = float(anneal)
        if apply:
            self.apply_gradients()

    def gradient_tensors(self):
        """the gradients of the last step(apply=False) (replicas.MultiVAEReplicas sums them over ranks)"""
        return list(self.G)

    def set_gradient_tensors(self, tensors):
        self.G = list(tensors)

    def apply_gradients(self):
        """the update half of a step: the learner on self.G (dW_q0, accumulated by row, is cleared behind it)"""
        tensors = [(self.params[k], self.M[k], self.V[k], self.G[k], k == 0) for k in range(len(self.params))]
        if self.learner is None:
            E.adam_dense_multi(tensors, self.adam)
        else:
            self.learner.apply(tensors)
        self.adam.advance()
        self.t += 1

    def loss(self):
        neg_ll, kl = (float(x) for x in self.stats.cpu())
        reg_var = self.reg * float(self.regsum.item()) / 2.0 if self.reg != 0.0 else 0.0
        return neg_ll + self.last_anneal * kl + 2.0 * reg_var, neg_ll, kl
