"""NGCF (alg_type = ngcf) for ANY embedding_size / layer_size (model/general_recommender/NGCF.py:31-33,160-202,271-286
build layers of whatever widths the config names; conf/NGCF.properties ships 16 / [16, 16], the NGCF paper uses
64 / [64, 64, 64]).  trainer.NGCFEngine is the fused, register-resident form for the shipped width; this engine
strings the general pieces into the same step:

  layer forward   S = A_hat E (SpMM) -> T1 = S W_gc + b_gc, T2 = (E .* S) W_bi + b_bi on the fp32 matrix cores
                  (csrc/gemm.hip: S and E .* S row-major, the weights [in][out], each read as it lies) -> leaky_relu sum,
                  dropout, l2_normalize
                  (csrc/ngcf_wide.hip) -> this layer's column block of the concatenated output
  head            the BPR head of NGCF.py:91-110 on rows of the concatenated output (the BPR-MF head kernel)
  layer backward  dT1, dT2 row-wise -> dW = S^T dT1, (E .* S)^T dT2 (contractions over the N node rows, split and
                  added in order), db = column sums -> dT W^T through the GEMM -> dS, the direct dE -> A_hat^T dS (SpMM)
  update          TF's dense ApplyAdam on the ego embeddings and on every layer weight (NGCF.py:112-114)

Buffers the SpMM touches are padded to its row widths (16 / 32 / 64 / 128 / 256) with zero columns; the products run
on the real widths.  Pinned to the reference class at 64 / [64, 64, 64] and 24 / [32, 8]
(tests/golden/tfgraph_ngcf_wide_*.npz)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import engine as E
from ._lib import call
from .engine import _ptr, _stream

_SPMM_WIDTHS = (16, 32, 64, 128, 256)


def _pad(w):
    for p in _SPMM_WIDTHS:
        if w <= p:
            return p
    raise NotImplementedError("NGCF layer width %d > 256 is not built" % w)


class NGCFWideEngine:
    def __init__(self, adj, adj_t, n_users, n_items, embed, weights, lr, reg, mess_dropout, max_batch, seed=2017,
                 learner="adam"):
        dev = E.require_gpu()
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.N = N = self.n_users + self.n_items
        self.A = E.SpmmCSR.from_scipy(adj, split_row=n_users)
        self.At = E.SpmmCSR.from_scipy(adj_t, split_row=n_users)
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        embed = np.asarray(embed, np.float32)
        self.L = len(weights)
        self.w = [embed.shape[1]] + [int(np.shape(ws[0])[1]) for ws in weights]
        for k, ws in enumerate(weights):
            assert tuple(np.shape(ws[0])) == (self.w[k], self.w[k + 1]) == tuple(np.shape(ws[2])), "W_gc / W_bi shapes"
        self.wp = [_pad(x) for x in self.w]
        self.d = self.w[0]
        self.dsum = sum(self.w)
        if self.dsum > 256:
            raise NotImplementedError("NGCF: concatenated output width %d > 256 is not built (BPR head rows)" % self.dsum)
        self.off = np.concatenate([[0], np.cumsum(self.w)]).astype(int)
        # the ego embeddings are kept padded to the SpMM's row width; pad columns stay zero under Adam
        self.E0p = z(N, self.wp[0])
        self.E0p[:, :self.w[0]] = f(embed)
        self.W = [tuple(f(np.reshape(x, -1) if j % 2 else x) for j, x in enumerate(ws)) for ws in weights]
        self.Out, self.dOut = z(N, self.dsum), z(N, self.dsum)
        self.ego = [self.E0p] + [z(N, self.wp[k + 1]) for k in range(self.L)]
        self.S = [z(N, self.wp[k]) for k in range(self.L)]
        self.X2 = [z(N, self.wp[k]) for k in range(self.L)]
        self.T1 = [z(N, self.w[k + 1]) for k in range(self.L)]
        self.T2 = [z(N, self.w[k + 1]) for k in range(self.L)]
        self.mask = [torch.zeros(N, self.w[k + 1], dtype=torch.uint8, device=dev) for k in range(self.L)]
        wmax = max(self.w)
        self.dT1, self.dT2, self.Y1, self.Y2 = (z(N * wmax) for _ in range(4))
        self.dS = [z(N, p) for p in sorted(set(self.wp))]
        self.dEd = [z(N, p) for p in sorted(set(self.wp))]
        self.dEgo = [[z(N, p) for p in sorted(set(self.wp))] for _ in range(2)]
        self.gE0 = z(N, self.wp[0])
        self.mE, self.vE = z(N, self.wp[0]), z(N, self.wp[0])
        self.gW = [tuple(torch.zeros_like(x) for x in ws) for ws in self.W]
        self.mW = [tuple(torch.zeros_like(x) for x in ws) for ws in self.W]
        self.vW = [tuple(torch.zeros_like(x) for x in ws) for ws in self.W]
        self.keep = 1.0 - float(mess_dropout)
        self.reg, self.seed, self.t = float(reg), int(seed), 0
        self.adam = E.AdamState(lr)
        self.terms = torch.empty(8 * max_batch, dtype=torch.float32, device=dev)
        self.rows = torch.zeros(3 * max_batch, dtype=torch.int32, device=dev)
        self.flag = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.max_batch = max_batch
        self.cs_ws = z(((N + 511) // 512) * wmax)                   # chunk sums of the bias gradients
        self.splits = max(1, min(768, (N + 255) // 256))             # cuts of the N-long contractions of dW: from N
        nbytes = C.c_size_t(0)
        call("nrhip_gemm_workspace_bytes", wmax, wmax, self.splits, C.byref(nbytes))
        self.ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=dev)
        # learner.py:2-17: adam inside the native step; the other learners after the spelled-out step's gradients
        self.learner = E.make_learner(learner, lr)
        if self.learner is not None:
            self.learner.init_slots([self.mE] + [m for ms in self.mW for m in ms], [self.vE] + [v for vs in self.vW for v in vs])
        self._native = self._native_buffers() if (self.L <= _lib.NGCF_WIDE_MAX_LAYERS and self.learner is None) else None

    def _native_buffers(self):
        """nrhip_ngcf_wide_buffers: every pointer the native step needs, recorded once (the buffers never move)"""
        b = _lib.NGCFWideBuffers()
        ptr = lambda t: t.data_ptr()
        for A, sfx, ws_name in ((self.A, "", "ws_fwd"), (self.At, "_t", "ws_bwd")):
            setattr(b, "plan" + sfx, A.plan.value)
            setattr(b, "indptr" + sfx, ptr(A.indptr))
            setattr(b, "indices" + sfx, ptr(A.indices))
            setattr(b, "vals" + sfx, ptr(A.vals))
            for k in range(self.L):
                A.ensure_schedule(self.wp[k])                        # the lane-group schedule of this width, attached once
                ws = A._workspace(self.wp[k])
                getattr(b, ws_name)[k] = ptr(ws)
                getattr(b, ws_name + "_bytes")[k] = ws.numel()
        b.n_users, b.n_nodes, b.n_layers, b.max_batch, b.dsum, b.splits = \
            self.n_users, self.N, self.L, self.max_batch, self.dsum, self.splits
        for k in range(self.L + 1):
            b.w[k], b.wp[k] = self.w[k], self.wp[k]
            b.ego[k] = ptr(self.ego[k])
        for k in range(self.L + 2):
            b.off[k] = int(self.off[k])
        for name in ("E0p", "mE", "vE", "gE0", "Out", "dOut", "dT1", "dT2", "Y1", "Y2", "terms", "cs_ws", "rows", "flag"):
            setattr(b, name, ptr(getattr(self, name)))
        b.cs_ws_bytes = self.cs_ws.numel() * 4
        b.gemm_ws, b.gemm_ws_bytes = ptr(self.ws), self.ws.numel()
        for k in range(self.L):
            b.S[k], b.X2[k], b.T1[k], b.T2[k], b.mask[k] = (ptr(x[k]) for x in (self.S, self.X2, self.T1, self.T2, self.mask))
            pi = self.wp[k]
            b.dS[k], b.dEd[k] = ptr(self._buf(self.dS, pi)), ptr(self._buf(self.dEd, pi))
            b.dEgo[k] = ptr(self._buf(self.dEgo[k % 2], pi))
            for j in range(4):
                b.W[k][j], b.gW[k][j], b.mW[k][j], b.vW[k][j] = (ptr(x[k][j]) for x in (self.W, self.gW, self.mW, self.vW))
        b.reg, b.keep = self.reg, self.keep
        return b

    # the trainable ego table as the caller sees it (real width)
    @property
    def E0(self):
        return self.E0p[:, :self.w[0]]

    def _buf(self, group, width):
        return group[sorted(set(self.wp)).index(width)]

    def _gemm(self, A, lda, a_kminor, Bm, ldb, b_kminor, M, N, K, Cm, ldc, splits=1, bias=None):
        call("nrhip_gemm_f32", _ptr(A), int(lda), int(a_kminor), _ptr(Bm), int(ldb), int(b_kminor), int(M), int(N),
             int(K), _ptr(Cm), int(ldc), 0, _ptr(bias, torch.float32, allow_none=True), -1, int(splits), _ptr(self.ws),
             self.ws.numel() if splits > 1 else 0, _stream())

    def forward(self, masks=None):
        """Fills self.Out = concat(E0, out_1 .. out_L) (NGCF.py:160-202).  masks: optional list of uint8 [N][w_k]
        device tensors (tests); otherwise a fresh dropout draw per call — evaluation included, as in the reference
        (NGCF.py:193 has no training flag).  One native call (nrhip_ngcf_wide_forward); forward_reference is the same
        launch sequence issued from Python."""
        if self._native is None:
            return self.forward_reference(masks)
        if masks is not None:
            for k in range(self.L):
                self.mask[k].copy_(masks[k])
        call("nrhip_ngcf_wide_forward", C.byref(self._native), 1 if masks is not None else 0,
             C.c_uint64(self.seed & (2**64 - 1)), C.c_uint64(self.t), _stream())
        self.t += 1
        return self.Out

    def forward_reference(self, masks=None):
        N = self.N
        E.copy2d(self.E0, self.Out[:, :self.w[0]])
        for k in range(self.L):
            wi, wo, pi, po = self.w[k], self.w[k + 1], self.wp[k], self.wp[k + 1]
            ego, S, X2 = self.ego[k], self.S[k], self.X2[k]
            self.A.matmul(ego, out=S)
            call("nrhip_ew_mul", _ptr(ego), pi, _ptr(S), pi, N, wi, _ptr(X2), pi, _stream())
            Wg, bg, Wb, bb = self.W[k]
            self._gemm(S, pi, 1, Wg, wo, 0, N, wo, wi, self.T1[k], wo, bias=bg)
            self._gemm(X2, pi, 1, Wb, wo, 0, N, wo, wi, self.T2[k], wo, bias=bb)
            if masks is not None:
                self.mask[k].copy_(masks[k])
            out_block = self.Out[:, self.off[k + 1]:self.off[k + 2]]
            call("nrhip_ngcf_act_fwd", _ptr(self.T1[k]), _ptr(self.T2[k]), wo, N, wo, po, float(self.keep),
                 _ptr(self.mask[k], torch.uint8), 1 if masks is not None else 0, C.c_uint64(self.seed & (2**64 - 1)),
                 C.c_uint64(self.t), k, _ptr(self.ego[k + 1]), po, C.c_void_p(out_block.data_ptr()), self.dsum,
                 _stream())
        self.t += 1
        return self.Out

    def final_embeddings(self):
        out = self.forward()
        return out[:self.n_users], out[self.n_users:]

    def step(self, users, pos, neg, loss_out, masks=None, plan=None):
        """One optimiser step: ONE native call (nrhip_ngcf_wide_step enqueues the ~25 + 15 L launches; issued from
        Python they cost more host time than GPU time).  step_reference spells the sequence out."""
        if self._native is None:
            return self.step_reference(users, pos, neg, loss_out, masks, plan)
        B = users.numel()
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        if masks is not None:
            for k in range(self.L):
                self.mask[k].copy_(masks[k])
        st = self.adam
        call("nrhip_ngcf_wide_step", C.byref(self._native), _ptr(users, torch.int32), _ptr(pos, torch.int32),
             _ptr(neg, torch.int32), B, _ptr(plan, torch.int64, allow_none=True), 1 if masks is not None else 0,
             C.c_uint64(self.seed & (2**64 - 1)), C.c_uint64(self.t), st.alpha(), st.beta1, st.beta2, st.eps,
             _ptr(loss_out, torch.float32, allow_none=True), _stream())
        self.t += 1
        self.adam.advance()

    def step_reference(self, users, pos, neg, loss_out, masks=None, plan=None):
        B, N, U = users.numel(), self.N, self.n_users
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        self.forward_reference(masks)
        rows = self.rows[:3 * B]
        E.lightgcn_mark_batch(users, pos, neg, U, rows, self.flag)
        E.bpr_mf_grad(self.Out[:U], self.Out[U:], users, pos, neg, self.reg, self.dOut[:U], self.dOut[U:], self.terms,
                      loss_out, plan)
        dego = None
        for k in range(self.L - 1, -1, -1):
            wi, wo, pi, po = self.w[k], self.w[k + 1], self.wp[k], self.wp[k + 1]
            Wg, bg, Wb, bb = self.W[k]
            gWg, gbg, gWb, gbb = self.gW[k]
            dblock = self.dOut[:, self.off[k + 1]:self.off[k + 2]]
            call("nrhip_ngcf_act_bwd", C.c_void_p(dblock.data_ptr()), self.dsum, _ptr(dego, allow_none=True), po,
                 _ptr(self.ego[k + 1]), po, _ptr(self.T1[k]), _ptr(self.T2[k]), wo, _ptr(self.mask[k], torch.uint8), N,
                 wo, float(self.keep), _ptr(self.dT1), _ptr(self.dT2), _stream())
            # weight gradients: contractions over the N rows, both operands k-major as stored
            self._gemm(self.S[k], pi, 0, self.dT1, wo, 0, wi, wo, N, gWg, wo, splits=self.splits)
            self._gemm(self.X2[k], pi, 0, self.dT2, wo, 0, wi, wo, N, gWb, wo, splits=self.splits)
            call("nrhip_colsum_rows", _ptr(self.dT1), wo, N, wo, _ptr(gbg), _ptr(self.cs_ws), self.cs_ws.numel() * 4, _stream())
            call("nrhip_colsum_rows", _ptr(self.dT2), wo, N, wo, _ptr(gbb), _ptr(self.cs_ws), self.cs_ws.numel() * 4, _stream())
            # Y1 = dT1 W_gc^T, Y2 = dT2 W_bi^T: both operands k-minor as stored
            self._gemm(self.dT1, wo, 1, Wg, wo, 1, N, wi, wo, self.Y1, wi)
            self._gemm(self.dT2, wo, 1, Wb, wo, 1, N, wi, wo, self.Y2, wi)
            dS, dEd = self._buf(self.dS, pi), self._buf(self.dEd, pi)
            call("nrhip_ngcf_mix_bwd", _ptr(self.Y1), _ptr(self.Y2), wi, _ptr(self.ego[k]), _ptr(self.S[k]), pi, N, wi,
                 pi, _ptr(dS), _ptr(dEd), _stream())
            nxt = self._buf(self.dEgo[k % 2], pi)
            self.At.matmul(dS, out=nxt, addend=dEd)                  # dE_k = dBi .* S + A_hat^T dS
            dego = nxt
        w0 = self.w[0]
        if dego is None:
            E.copy2d(self.dOut[:, :w0], self.gE0[:, :w0])
        else:
            E.add2d(self.dOut[:, :w0], dego[:, :w0], self.gE0[:, :w0])
        tensors = [(self.E0p, self.mE, self.vE, self.gE0)] + \
            [(w, m, v, g) for k in range(self.L) for w, m, v, g in zip(self.W[k], self.mW[k], self.vW[k], self.gW[k])]
        if self.learner is None:
            E.adam_dense_multi(tensors, self.adam)
        else:
            self.learner.apply(tensors)
        E.rows_clear(rows, self.dsum, (self.dOut,), self.flag)
        self.adam.advance()
