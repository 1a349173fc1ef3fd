"""Row-sharded NGCF over the GPUs of one node (SURVEY §8e last row: "NGCF as LightGCN + replicated tiny W").

The reference splits Â into 100 row slabs to fit ONE device (NGCF.py:320-332) — numerically a single SpMM.  Here the
rows are split across ranks for real (parallel.BipartitePartition, as sharded.ShardedLightGCN): every rank owns a
slice of the users and of the items — their ego embeddings with the Adam moments, every per-layer buffer, and the rows
of Â (`norm`: D⁻¹(A + I), not symmetric) and of Âᵀ for those nodes.  The layer weights (W_gc, b_gc, W_bi, b_bi:
a few KB) are replicated.  One step (NGCF.py:91-114,160-202):

  layer forward   all-gather of the layer input [b][w] -> S = Â_block · X (SpMM) -> T1, T2, activation, dropout,
                  l2_normalize on the rank's rows (ngcf_wide.py's kernels on b rows instead of N)
  head            the batch's rows of the concatenated output live on their owners: ids -> owners -> rows ->
                  BPR head on the compact [3B][Σw] block -> gradient rows back, added at the owners in the order of
                  the GLOBAL batch (sharded.RowRouter: no atomics)
  layer backward  row-wise on the rank's rows; dW / db are contractions over the node rows: every rank contracts its
                  rows, ONE all-reduce per step sums the (tiny) weight gradients; dS crosses ranks through the
                  all-gather + Âᵀ_block SpMM
  update          dense TF-Adam: the rank's ego rows; the replicated weights (identical on every rank: same
                  initial values, same all-reduced gradients)

What is exact: every node row sees the single-GPU arithmetic (same kernels; a row's neighbour sum runs in ascending
node-id order).  The weight gradients are sums over ALL nodes: per-rank partial sums + an all-reduce associate
differently from one N-long contraction — last-ulp differences in dW, inside north_star's 1e-5
(tests/test_sharded_ngcf_gpu.py: two ranks against NGCFWideEngine).  Dropout masks are drawn per (row, column) of the
rank's block: a different (equally distributed) draw from the single-GPU engine's unless masks are handed in.
"""
import ctypes as C

import numpy as np
import torch

from . import engine as E
from . import parallel
from ._lib import call
from .engine import _ptr, _stream
from .ngcf_wide import _pad
from .sharded import RowRouter, _compact_neg


def _local_block(part, rank, a):
    """rows [users of `rank`; items of `rank`] of the scipy CSR `a`, padded to part.b rows, columns = positions in the
    rank-major gathered layout; storage order = ascending node id (the order the row sums run in)"""
    import scipy.sparse as sp
    (ulo, uhi), (ilo, ihi) = part.users_of(rank), part.items_of(rank)
    nu, ni, bu, b = uhi - ulo, ihi - ilo, part.bu, part.b
    blk = sp.vstack([a[ulo:uhi], a[part.U + ilo:part.U + ihi]]).tocsr()
    ip = np.zeros(b + 1, dtype=np.int64)
    ip[1:nu + 1] = blk.indptr[1:nu + 1]
    ip[nu + 1:bu + 1] = blk.indptr[nu]
    ip[bu + 1:bu + ni + 1] = blk.indptr[nu + 1:nu + ni + 1]
    ip[bu + ni + 1:] = blk.indptr[nu + ni]
    cols = part.position(blk.indices.astype(np.int64)).astype(np.int32)
    return E.SpmmCSR(ip, cols, blk.data.astype(np.float32), n_cols=part.n_pad)


class ShardedNGCF:
    def __init__(self, comm, adj, adj_t, n_users, n_items, embed, weights, lr, reg, mess_dropout, max_batch, seed=2017):
        dev = E.require_gpu()
        self.comm, self.rank, self.world = comm, comm.rank, comm.world
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.part = part = parallel.BipartitePartition(self.n_users, self.n_items, self.world)
        self.b, self.Npad = part.b, part.n_pad
        a = adj.tocsr().astype(np.float32)
        a.sort_indices()
        at = adj_t.tocsr().astype(np.float32)
        at.sort_indices()
        self.A, self.At = _local_block(part, self.rank, a), _local_block(part, self.rank, at)
        f = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(dev)
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        embed = np.asarray(embed, np.float32)
        self.L = len(weights)
        self.w = [embed.shape[1]] + [int(np.shape(ws[0])[1]) for ws in weights]
        self.wp = [_pad(x) for x in self.w]
        self.dsum = sum(self.w)
        if self.dsum > 256:
            raise NotImplementedError("NGCF: concatenated output width %d > 256 is not built (BPR head rows)" % self.dsum)
        self.off = np.concatenate([[0], np.cumsum(self.w)]).astype(int)
        (ulo, uhi), (ilo, ihi) = part.users_of(self.rank), part.items_of(self.rank)
        b = self.b
        self.E0p = z(b, self.wp[0])                                   # ego rows of this rank, padded to the SpMM width
        self.E0p[:uhi - ulo, :self.w[0]] = f(embed[ulo:uhi])
        self.E0p[part.bu:part.bu + ihi - ilo, :self.w[0]] = f(embed[self.n_users + ilo:self.n_users + ihi])
        self.W = [tuple(f(np.reshape(x, -1) if j % 2 else x) for j, x in enumerate(ws)) for ws in weights]
        self.Out, self.dOut = z(b, self.dsum), z(b, self.dsum)
        self.ego = [self.E0p] + [z(b, self.wp[k + 1]) for k in range(self.L)]
        self.S = [z(b, self.wp[k]) for k in range(self.L)]
        self.X2 = [z(b, self.wp[k]) for k in range(self.L)]
        self.T1 = [z(b, self.w[k + 1]) for k in range(self.L)]
        self.T2 = [z(b, self.w[k + 1]) for k in range(self.L)]
        self.mask = [torch.zeros(b, self.w[k + 1], dtype=torch.uint8, device=dev) for k in range(self.L)]
        wmax = max(self.w)
        self.dT1, self.dT2, self.Y1, self.Y2 = (z(b * wmax) for _ in range(4))
        widths = sorted(set(self.wp))
        self._widths = widths
        self.dS = [z(b, p) for p in widths]
        self.dEd = [z(b, p) for p in widths]
        self.dEgo = [[z(b, p) for p in widths] for _ in range(2)]
        self.Xg = [z(self.Npad, p) for p in widths]                   # gathered operand of a hop, per width
        self.gE0 = z(b, self.wp[0])
        self.mE, self.vE = z(b, self.wp[0]), z(b, self.wp[0])
        self.gW = [tuple(torch.zeros_like(x) for x in ws) for ws in self.W]
        self.mW = [tuple(torch.zeros_like(x) for x in ws) for ws in self.W]
        self.vW = [tuple(torch.zeros_like(x) for x in ws) for ws in self.W]
        self.keep = 1.0 - float(mess_dropout)
        self.reg, self.seed, self.t = float(reg), int(seed), 0
        self.adam = E.AdamState(lr)
        self.max_batch = int(max_batch)
        B3 = 3 * self.max_batch
        self.terms = torch.empty(8 * self.max_batch, dtype=torch.float32, device=dev)
        self.flag = torch.zeros(b, dtype=torch.uint8, device=dev)
        self.compact, self.gcompact = z(B3, self.dsum), z(B3, self.dsum)
        self._ar = torch.arange(self.max_batch, dtype=torch.int32, device=dev)
        self.router = RowRouter(comm, part, self.max_batch)
        self.cs_ws = z(((b + 511) // 512) * wmax)
        self.splits = max(1, min(768, (b + 255) // 256))              # cuts of the b-long contractions of dW
        nbytes = C.c_size_t(0)
        call("nrhip_gemm_workspace_bytes", wmax, wmax, self.splits, C.byref(nbytes))
        self.ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=dev)
        self._gidx = None

    def _buf(self, group, width):
        return group[self._widths.index(width)]

    def _gemm(self, A, lda, a_kminor, Bm, ldb, b_kminor, M, N, K, Cm, ldc, splits=1, bias=None):
        call("nrhip_gemm_f32", _ptr(A), int(lda), int(a_kminor), _ptr(Bm), int(ldb), int(b_kminor), int(M), int(N),
             int(K), _ptr(Cm), int(ldc), 0, _ptr(bias, torch.float32, allow_none=True), -1, int(splits), _ptr(self.ws),
             self.ws.numel() if splits > 1 else 0, _stream())

    def _gather(self, local):
        """every rank's [b][w] block, rank-major: the operand of a hop (one all-gather)"""
        X = self._buf(self.Xg, local.shape[1])
        self.comm.all_gather_rows(local, X)
        return X

    def _compact_plan(self, B):
        cache = self.__dict__.setdefault("_plans", {})
        if B not in cache:
            ar = self._ar[:B]
            cache[B] = E.bpr_plan(ar, ar, (ar + B).contiguous(), B, B).clone()
        return cache[B]

    def local_masks(self, masks):
        """per-layer masks given for all N nodes ([N][w] uint8, tests) -> this rank's padded block rows"""
        (ulo, uhi), (ilo, ihi) = self.part.users_of(self.rank), self.part.items_of(self.rank)
        out = []
        for m in masks:
            m = torch.as_tensor(m, dtype=torch.uint8)
            loc = torch.zeros((self.b, m.shape[1]), dtype=torch.uint8)
            loc[:uhi - ulo] = m[ulo:uhi]
            loc[self.part.bu:self.part.bu + ihi - ilo] = m[self.n_users + ilo:self.n_users + ihi]
            out.append(loc.to(self.E0p.device))
        return out

    # ------------------------------------------------------------------ forward (NGCF.py:160-202) on this rank's rows
    def forward(self, masks=None):
        b = self.b
        E.copy2d(self.E0p[:, :self.w[0]], self.Out[:, :self.w[0]])
        for k in range(self.L):
            wi, wo, pi, po = self.w[k], self.w[k + 1], self.wp[k], self.wp[k + 1]
            ego, S, X2 = self.ego[k], self.S[k], self.X2[k]
            self.A.matmul(self._gather(ego), out=S)
            call("nrhip_ew_mul", _ptr(ego), pi, _ptr(S), pi, b, wi, _ptr(X2), pi, _stream())
            Wg, bg, Wb, bb = self.W[k]
            self._gemm(S, pi, 1, Wg, wo, 0, b, wo, wi, self.T1[k], wo, bias=bg)
            self._gemm(X2, pi, 1, Wb, wo, 0, b, wo, wi, self.T2[k], wo, bias=bb)
            if masks is not None:
                self.mask[k].copy_(masks[k])
            out_block = self.Out[:, self.off[k + 1]:self.off[k + 2]]
            call("nrhip_ngcf_act_fwd", _ptr(self.T1[k]), _ptr(self.T2[k]), wo, b, wo, po, float(self.keep),
                 _ptr(self.mask[k], torch.uint8), 1 if masks is not None else 0,
                 C.c_uint64((self.seed + 7919 * self.rank) & (2**64 - 1)), C.c_uint64(self.t), k, _ptr(self.ego[k + 1]),
                 po, C.c_void_p(out_block.data_ptr()), self.dsum, _stream())
        self.t += 1
        return self.Out

    def final_embeddings(self, masks=None):
        """the full (user, item) concatenated tables on every rank: one all-gather (evaluation entrance)"""
        out = self.forward(masks)
        full = torch.empty((self.Npad, self.dsum), dtype=torch.float32, device=out.device)
        self.comm.all_gather_rows(out, full)
        if self._gidx is None:
            self._gidx = self.part.gathered_index(out.device)
        return full[self._gidx[0]], full[self._gidx[1]]

    def ego_table(self):
        """the trainable ego embeddings in natural id order, gathered (tests / checkpoints)"""
        full = torch.empty((self.Npad, self.wp[0]), dtype=torch.float32, device=self.E0p.device)
        self.comm.all_gather_rows(self.E0p, full)
        if self._gidx is None:
            self._gidx = self.part.gathered_index(full.device)
        return torch.cat([full[self._gidx[0]], full[self._gidx[1]]])[:, :self.w[0]]

    # ------------------------------------------------------------------ one optimiser step on this rank's B triplets
    def step(self, users, pos, neg, loss_out, masks=None):
        B, b, dsum = users.numel(), self.b, self.dsum
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        self.forward(masks)
        # head: rows of Out from their owners, BPR on the compact block, gradient rows back in global-batch order
        rt = self.router.request(users, pos, neg, self.n_users)
        rows = torch.empty((rt.asked.numel(), dsum), dtype=torch.float32, device=users.device)
        E.rows_gather(rt.asked, self.Out, rows)
        got, _ = self.comm.all_to_all_rows(rows, rt.recv_counts, rt.send_counts)
        comp, gcomp = self.compact[:3 * B], self.gcompact[:3 * B]
        E.rows_gather(rt.inv, got, comp)                                   # comp[p] = answer to request p
        ar = self._ar[:B]
        E.bpr_mf_grad(comp[:B], comp[B:], ar, ar, _compact_neg(self, B), self.reg, gcomp[:B], gcomp[B:], self.terms,
                      loss_out, self._compact_plan(B))
        back = torch.empty((3 * B, dsum), dtype=torch.float32, device=users.device)
        E.rows_gather(rt.order, gcomp, back)
        mine, _ = self.comm.all_to_all_rows(back, rt.send_counts, rt.recv_counts)
        keys, index_of_pos = self.router.ordered_keys(rt)
        E.rows_sum_sorted(keys, index_of_pos, mine, self.dOut)
        # layers backwards
        dego = None
        for k in range(self.L - 1, -1, -1):
            wi, wo, pi, po = self.w[k], self.w[k + 1], self.wp[k], self.wp[k + 1]
            Wg, bg, Wb, bb = self.W[k]
            gWg, gbg, gWb, gbb = self.gW[k]
            dblock = self.dOut[:, self.off[k + 1]:self.off[k + 2]]
            call("nrhip_ngcf_act_bwd", C.c_void_p(dblock.data_ptr()), dsum, _ptr(dego, allow_none=True), po,
                 _ptr(self.ego[k + 1]), po, _ptr(self.T1[k]), _ptr(self.T2[k]), wo, _ptr(self.mask[k], torch.uint8), b,
                 wo, float(self.keep), _ptr(self.dT1), _ptr(self.dT2), _stream())
            self._gemm(self.S[k], pi, 0, self.dT1, wo, 0, wi, wo, b, gWg, wo, splits=self.splits)
            self._gemm(self.X2[k], pi, 0, self.dT2, wo, 0, wi, wo, b, gWb, wo, splits=self.splits)
            call("nrhip_colsum_rows", _ptr(self.dT1), wo, b, wo, _ptr(gbg), _ptr(self.cs_ws), self.cs_ws.numel() * 4, _stream())
            call("nrhip_colsum_rows", _ptr(self.dT2), wo, b, wo, _ptr(gbb), _ptr(self.cs_ws), self.cs_ws.numel() * 4, _stream())
            self._gemm(self.dT1, wo, 1, Wg, wo, 1, b, wi, wo, self.Y1, wi)
            self._gemm(self.dT2, wo, 1, Wb, wo, 1, b, wi, wo, self.Y2, wi)
            dS, dEd = self._buf(self.dS, pi), self._buf(self.dEd, pi)
            call("nrhip_ngcf_mix_bwd", _ptr(self.Y1), _ptr(self.Y2), wi, _ptr(self.ego[k]), _ptr(self.S[k]), pi, b, wi,
                 pi, _ptr(dS), _ptr(dEd), _stream())
            nxt = self._buf(self.dEgo[k % 2], pi)
            self.At.matmul(self._gather(dS), out=nxt, addend=dEd)          # dE_k = dBi .* S + Â^T dS
            dego = nxt
        w0 = self.w[0]
        if dego is None:
            E.copy2d(self.dOut[:, :w0], self.gE0[:, :w0])
        else:
            E.add2d(self.dOut[:, :w0], dego[:, :w0], self.gE0[:, :w0])
        # the replicated weights: every rank's partial contraction summed (the step's one all-reduce per tensor: KBs)
        for gws in self.gW:
            for g in gws:
                self.comm.allreduce_sum_(g)
        E.adam_dense_multi([(self.E0p, self.mE, self.vE, self.gE0)] +
                           [(w, m, v, g) for k in range(self.L)
                            for w, m, v, g in zip(self.W[k], self.mW[k], self.vW[k], self.gW[k])], self.adam)
        E.rows_clear(rt.asked, dsum, (self.dOut,), None)
        self.adam.advance()
