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
    ALGS = ("ngcf", "gcn", "gcmc")

    def __init__(self, adj, adj_t, n_users, n_items, embed, weights, lr, reg, mess_dropout, max_batch, seed=2017,
                 learner="adam", alg_type="ngcf", node_dropout=0.0):
        """alg_type (NGCF.py:67-74): "ngcf" (the layer of the module docstring); "gcn" (NGCF.py:204-224: a layer is
        dropout(leaky_relu(S W_gc + b_gc)), no bi-interaction, no normalisation; weights' slots 2, 3 are unused);
        "gcmc" (NGCF.py:226-248: conv = leaky_relu(S W_gc + b_gc) feeds the next layer, the output block is
        dropout(conv W_mlp + b_mlp) with (W_mlp, b_mlp) in slots 2, 3, and E0 is NOT part of the output).
        node_dropout (ngcf only, NGCF.py:162-164,334-362): every step (and every evaluation forward) drops each stored
        entry of the adjacency with this probability and scales the rest by 1 / keep."""
        dev = E.require_gpu()
        if alg_type not in self.ALGS:
            raise ValueError("alg_type must be one of %s" % (self.ALGS,))
        self.alg = alg_type
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
            assert tuple(np.shape(ws[0])) == (self.w[k], self.w[k + 1]), "W_gc shape"
            if self.alg != "gcn":
                assert tuple(np.shape(ws[2])) == (self.w[k], self.w[k + 1]), "W_bi / W_mlp shape"
            if self.alg == "gcmc" and self.w[k] != self.w[k + 1]:
                # NGCF.py:281-282 creates W_mlp_k as [w_k][w_{k+1}] and NGCF.py:241 multiplies a [N][w_{k+1}] matrix by it
                raise ValueError("alg_type=gcmc needs equal layer widths (W_mlp_%d is [%d][%d] and multiplies %d columns)"
                                 % (k, self.w[k], self.w[k + 1], self.w[k + 1]))
        self.wp = [_pad(x) for x in self.w]
        self.d = self.w[0]
        # column offsets of the concatenated output: block k + 1 is layer k's; gcmc leaves E0 out (a zero-width block 0)
        blocks = ([0] if self.alg == "gcmc" else [self.w[0]]) + self.w[1:]
        self.dsum = sum(blocks)
        if self.dsum > 256:
            raise NotImplementedError("NGCF: concatenated output width %d > 256 is not built (BPR head rows)" % self.dsum)
        self.off = np.concatenate([[0], np.cumsum(blocks)]).astype(int)
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
        self._native = self._native_buffers() if (self.L <= _lib.NGCF_WIDE_MAX_LAYERS and self.learner is None and
                                                  self.alg == "ngcf") else None
        self.node_keep = 1.0 - float(node_dropout)
        if self.alg != "ngcf":
            self.node_keep = 1.0                                      # gcn / gcmc never drop nodes (NGCF.py:205,227)
        if self.node_keep < 1.0:
            import scipy.sparse as sp
            a = adj.tocsr().astype(np.float32)
            a.sort_indices()
            pos = sp.csr_matrix((np.arange(1, a.nnz + 1, dtype=np.float64), a.indices, a.indptr), shape=a.shape)
            pt = pos.T.tocsr()
            pt.sort_indices()
            # entry j of the transposed matrix is entry t_of[j] of the matrix: both take the same draw
            self._t_of = torch.from_numpy((pt.data - 1).astype(np.int32)).to(dev)
            self._vals0 = self.A.vals.clone()
            self.edge_keep = torch.ones(max(self.A.nnz, 1), dtype=torch.uint8, device=dev)

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

    def forward(self, masks=None, node_keep_given=None):
        """Fills self.Out = concat(E0, out_1 .. out_L) (NGCF.py:160-202).  masks: optional list of uint8 [N][w_k]
        device tensors (tests); otherwise a fresh dropout draw per call — evaluation included, as in the reference
        (NGCF.py:193 has no training flag).  One native call (nrhip_ngcf_wide_forward); forward_reference is the same
        launch sequence issued from Python."""
        if node_keep_given is not None or self.node_keep < 1.0:
            self._drop_nodes(node_keep_given)
        if self.alg != "ngcf":
            return self._forward_variant(masks)
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

    def final_embeddings(self, masks=None, node_keep_given=None):
        out = self.forward(masks, node_keep_given)
        return out[:self.n_users], out[self.n_users:]

    def _drop_nodes(self, given=None):
        """NGCF.py:334-362: this step's (this forward's) draw over the adjacency's stored entries, the same draw for
        the transposed matrix; the lane-group schedules re-read the values (nrhip_spmm_blocked_pack)."""
        if self.node_keep >= 1.0 and given is None:
            return
        if given is not None:
            self.edge_keep.copy_(given)
        n = self.A.nnz
        call("nrhip_edge_dropout", _ptr(self._vals0), n, float(self.node_keep), _ptr(self.edge_keep, torch.uint8),
             1 if given is not None else 0, C.c_uint64(self.seed & (2**64 - 1)), C.c_uint64(self.t), _ptr(self.A.vals),
             _stream())
        call("nrhip_gather_f32", _ptr(self.A.vals), _ptr(self._t_of, torch.int32), n, _ptr(self.At.vals), _stream())
        self.A.values_changed()
        self.At.values_changed()

    def step(self, users, pos, neg, loss_out, masks=None, plan=None, node_keep_given=None):
        """One optimiser step: ONE native call (nrhip_ngcf_wide_step enqueues the ~25 + 15 L launches; issued from
        Python they cost more host time than GPU time).  step_reference spells the sequence out."""
        if self.alg != "ngcf":
            return self._step_variant(users, pos, neg, loss_out, masks, plan)
        if node_keep_given is not None or self.node_keep < 1.0:
            self._drop_nodes(node_keep_given)
        if self._native is None:
            return self.step_reference(users, pos, neg, loss_out, masks, plan, dropped=True)
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

    def step_reference(self, users, pos, neg, loss_out, masks=None, plan=None, dropped=False):
        B, N, U = users.numel(), self.N, self.n_users
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        if not dropped:
            self._drop_nodes()
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

    # ------------------------------------------------------------------ alg_type = gcn / gcmc (NGCF.py:204-248)
    def _act(self, T, wo, po, k, flags, out_a, out_b, masks_given):
        call("nrhip_lrelu_drop_fwd", _ptr(T), wo, self.N, wo, po, float(self.keep), _ptr(self.mask[k], torch.uint8),
             1 if masks_given else 0, C.c_uint64(self.seed & (2**64 - 1)), C.c_uint64(self.t), k, flags,
             _ptr(out_a, allow_none=True), po, None if out_b is None else C.c_void_p(out_b.data_ptr()), self.dsum, _stream())

    def _forward_variant(self, masks=None):
        N, gcmc = self.N, self.alg == "gcmc"
        if not gcmc:
            E.copy2d(self.E0, self.Out[:, :self.w[0]])
        for k in range(self.L):
            wi, wo, pi, po = self.w[k], self.w[k + 1], self.wp[k], self.wp[k + 1]
            Wg, bg, Wx, bx = self.W[k]
            self.A.matmul(self.ego[k], out=self.S[k])
            self._gemm(self.S[k], pi, 1, Wg, wo, 0, N, wo, wi, self.T1[k], wo, bias=bg)
            if masks is not None:
                self.mask[k].copy_(masks[k])
            block = self.Out[:, self.off[k + 1]:self.off[k + 2]]
            if not gcmc:     # gcn: E' = dropout(leaky_relu(T1)) is the next layer's input AND this layer's output block
                self._act(self.T1[k], wo, po, k, 3, self.ego[k + 1], block, masks is not None)
            else:            # gcmc: conv = leaky_relu(T1) goes on; the block is dropout(conv W_mlp + b_mlp)
                self._act(self.T1[k], wo, po, k, 1, self.ego[k + 1], None, False)
                self._gemm(self.ego[k + 1], po, 1, Wx, wo, 0, N, wo, wo, self.T2[k], wo, bias=bx)
                self._act(self.T2[k], wo, po, k, 2, None, block, masks is not None)
        self.t += 1
        return self.Out

    def _step_variant(self, users, pos, neg, loss_out, masks=None, plan=None):
        B, N, U, gcmc = users.numel(), self.N, self.n_users, self.alg == "gcmc"
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        self._forward_variant(masks)
        rows = self.rows[:3 * B]
        E.lightgcn_mark_batch(users, pos, neg, U, rows, self.flag)
        E.bpr_mf_grad(self.Out[:U], self.Out[U:], users, pos, neg, self.reg, self.dOut[:U], self.dOut[U:], self.terms,
                      loss_out, plan)
        bwd = lambda da, lda, db, ldb, T, k, flags, dT, wo: call(
            "nrhip_lrelu_drop_bwd", C.c_void_p(da.data_ptr()), lda, _ptr(db, allow_none=True), ldb, _ptr(T), wo,
            _ptr(self.mask[k], torch.uint8), N, wo, float(self.keep), flags, _ptr(dT), _stream())
        colsum = lambda src, wo, dst: call("nrhip_colsum_rows", _ptr(src), wo, N, wo, _ptr(dst), _ptr(self.cs_ws),
                                           self.cs_ws.numel() * 4, _stream())
        dego = None
        for k in range(self.L - 1, -1, -1):
            wi, wo, pi, po = self.w[k], self.w[k + 1], self.wp[k], self.wp[k + 1]
            Wg, bg, Wx, bx = self.W[k]
            gWg, gbg, gWx, gbx = self.gW[k]
            dblock = self.dOut[:, self.off[k + 1]:self.off[k + 2]]
            if not gcmc:
                bwd(dblock, self.dsum, dego, po, self.T1[k], k, 3, self.dT1, wo)
            else:
                bwd(dblock, self.dsum, None, 0, self.T2[k], k, 2, self.dT2, wo)          # through the dropout
                self._gemm(self.ego[k + 1], po, 0, self.dT2, wo, 0, wo, wo, N, gWx, wo, splits=self.splits)   # dW_mlp
                colsum(self.dT2, wo, gbx)
                self._gemm(self.dT2, wo, 1, Wx, wo, 1, N, wo, wo, self.Y2, wo)           # d conv (through the dense layer)
                y2 = self.Y2[:N * wo].view(N, wo)
                bwd(y2, wo, dego, po, self.T1[k], k, 1, self.dT1, wo)                    # + what the next layer sends back
            self._gemm(self.S[k], pi, 0, self.dT1, wo, 0, wi, wo, N, gWg, wo, splits=self.splits)
            colsum(self.dT1, wo, gbg)
            self._gemm(self.dT1, wo, 1, Wg, wo, 1, N, wi, wo, self.Y1, wi)
            dS = self._buf(self.dS, pi)
            E.copy2d(self.Y1[:N * wi].view(N, wi), dS[:, :wi])                           # pad columns stay zero
            nxt = self._buf(self.dEgo[k % 2], pi)
            self.At.matmul(dS, out=nxt)                                                  # dE_k = A_hat^T dS
            dego = nxt
        w0 = self.w[0]
        if gcmc:
            E.copy2d(dego[:, :w0], self.gE0[:, :w0])                                     # E0 reaches the loss through layer 0 only
        elif dego is None:
            E.copy2d(self.dOut[:, :w0], self.gE0[:, :w0])
        else:
            E.add2d(self.dOut[:, :w0], dego[:, :w0], self.gE0[:, :w0])
        used = (0, 1) if not gcmc else (0, 1, 2, 3)
        tensors = [(self.E0p, self.mE, self.vE, self.gE0)] + \
            [(self.W[k][j], self.mW[k][j], self.vW[k][j], self.gW[k][j]) for k in range(self.L) for j in used]
        if self.learner is None:
            E.adam_dense_multi(tensors, self.adam)
        else:
            self.learner.apply(tensors)
        E.rows_clear(rows, self.dsum, (self.dOut,), self.flag)
        self.adam.advance()
