"""Row-sharded LightGCN / BPR-MF over the GPUs of one node (BASELINE config 4: tables that are not
meant to be replicated — U = 10⁷, I = 10⁶, d = 128 is 5.6 GB per [N][d] buffer and the step keeps
a dozen of them).

Partition: the N = U + I node rows are cut into `world` padded blocks of b = ⌈N/world⌉ rows; rank r
owns rows [r·b, (r+1)·b) of the embedding table E0, of its Adam moments and of every layer
buffer, plus the CSR rows of Â (and of Âᵀ when Â is not symmetric) for those nodes.

One LightGCN step (LightGCN.py:132-166,178) then has exactly these exchange points:
  * per propagation hop (L forward, L backward): ONE all-gather of the [b][d] blocks into the
    [world·b][d] operand of the local SpMM  (RCCL all-gather over xGMI; N·d·4 bytes per hop);
  * BPR head: the 3·B rows a rank's triplets touch live on their owners — ids go out with an
    all-to-all, owners answer with the rows (Esum and E0 side by side), the head runs locally on
    the compact [3B][d] block, and the 3·B gradient rows return by the reverse all-to-all and are
    scatter-added into the owners' buffers;
  * Adam is owner-local (dense TF-Adam on the rank's block): no exchange.
The result equals the single-process step on the concatenated global batch (tests: two ranks
vs one process).  All arithmetic is the same HIP kernels as the replicated engine; the
collectives are torch.distributed plumbing (`parallel.Comm`).
"""
import numpy as np
import torch

from . import engine as E
from . import parallel


class ShardedLightGCN:
    def __init__(self, comm, adj_csr, n_users, n_items, embed, n_layers, lr, reg, max_batch,
                 symmetric=None):
        dev = E.require_gpu()
        self.comm, self.rank, self.world = comm, comm.rank, comm.world
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.N = self.n_users + self.n_items
        self.L, self.reg, self.max_batch = int(n_layers), float(reg), int(max_batch)
        self.b = parallel.block_size(self.N, self.world)
        self.lo = min(self.rank * self.b, self.N)
        self.hi = min(self.lo + self.b, self.N)
        self.n_loc = self.hi - self.lo
        self.Npad = self.b * self.world
        a = adj_csr.tocsr().astype(np.float32)
        a.sort_indices()
        if symmetric is None:
            symmetric = (a != a.T).nnz == 0
        # local row block, padded to b rows (empty rows) so every rank runs the same shapes
        self.A = self._local_rows(a)
        self.At = self.A if symmetric else self._local_rows(a.T.tocsr())
        embed = np.asarray(embed, dtype=np.float32)
        self.d = embed.shape[1]
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        self.E0 = z(self.b, self.d)
        self.E0[:self.n_loc] = torch.from_numpy(np.ascontiguousarray(embed[self.lo:self.hi])).to(dev)
        self.m, self.v = z(self.b, self.d), z(self.b, self.d)
        self.X = z(self.Npad, self.d)                       # gathered operand of the local SpMM
        self.Ya, self.Yb, self.Esum = (z(self.b, self.d) for _ in range(3))
        self.H, self.Greg, self.Ga, self.Gb = (z(self.b, self.d) for _ in range(4))
        B3 = 3 * self.max_batch
        self.req_rows = z(B3, 2 * self.d)                   # [Esum | E0] rows of my triplets
        self.gc_star, self.gc_reg = z(B3, self.d), z(B3, self.d)
        self.terms = z(8 * self.max_batch)
        self.adam = E.AdamState(lr)
        self._cu = torch.arange(self.max_batch, dtype=torch.int32, device=dev)
        self._cp = self._cu.clone()

    def _local_rows(self, a):
        blk = a[self.lo:self.hi]
        indptr = np.zeros(self.b + 1, dtype=np.int64)
        indptr[1:self.n_loc + 1] = blk.indptr[1:]
        indptr[self.n_loc + 1:] = blk.indptr[-1]
        return E.SpmmCSR(indptr, blk.indices, blk.data, n_cols=self.Npad)

    # ------------------------------------------------------------------ propagation
    def propagate(self):
        """Esum (this rank's rows) = Σ_k E^k; returns it (E* = Esum / (L+1))."""
        if self.L == 0:
            self.Esum.copy_(self.E0)
            return self.Esum
        src, acc_in = self.E0, self.E0
        ping = (self.Ya, self.Yb)
        for k in range(self.L):
            self.comm.all_gather_rows(src, self.X)                      # exchange: one all-gather per hop
            out = ping[k & 1]
            self.A.matmul(self.X, out=out, sum_in=acc_in, sum_out=self.Esum)
            src, acc_in = out, self.Esum
        return self.Esum

    def final_embeddings_local(self):
        out = torch.empty_like(self.Esum)
        E.div_scalar(self.propagate(), float(self.L + 1), out)
        return out[:self.n_loc]

    def final_embeddings(self):
        """Full (user, item) tables on every rank (one all-gather; evaluation entrance)."""
        loc = torch.zeros_like(self.Esum)
        E.div_scalar(self.propagate(), float(self.L + 1), loc)
        self.comm.all_gather_rows(loc, self.X)
        return self.X[:self.n_users], self.X[self.n_users:self.N]

    # ------------------------------------------------------------------ lookups
    def _route(self, node_ids):
        """Sort requested global node ids by owner: (order, local ids sorted, counts)."""
        owner = torch.div(node_ids, self.b, rounding_mode="floor")
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=self.world)[:self.world].cpu().tolist()
        local = (node_ids - owner * self.b)[order].to(torch.int32).contiguous()
        return order, local, counts

    def step(self, users, pos, neg, loss_out=None):
        """One optimiser step on this rank's B triplets (global batch = all ranks' triplets)."""
        B, d, dev = users.numel(), self.d, self.E0.device
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        esum = self.propagate()
        # --- lookup: ids -> owners -> [Esum | E0] rows back
        nodes = torch.cat([users.long(), pos.long() + self.n_users, neg.long() + self.n_users])
        order, local, counts = self._route(nodes)
        asked, asked_counts = self.comm.all_to_all_rows(local, counts)
        rows = torch.empty((asked.numel(), 2 * d), dtype=torch.float32, device=dev)
        E.rows_gather(asked, esum, rows[:, :d])
        E.rows_gather(asked, self.E0, rows[:, d:])
        got, _ = self.comm.all_to_all_rows(rows, asked_counts)
        req = self.req_rows[:3 * B]
        req[order] = got                                               # back to request order (plumbing copy)
        es = req[:, :d].contiguous()
        e0 = req[:, d:].contiguous()
        # --- BPR head on the compact block: "users" = rows [0,B), "items" = rows [B,3B)
        gs, gr = self.gc_star[:3 * B], self.gc_reg[:3 * B]
        E.lightgcn_bpr_grad(es, e0, B, self.L, self._cu[:B], self._cp[:B],
                            (self._cp[:B] + B).contiguous(), self.reg, gs, gr, self.terms, loss_out)
        # --- gradients back to the owners, summed into H (= dL/dE*, then /(L+1)) and Greg
        back = torch.cat([gs, gr], dim=1)[order].contiguous()
        mine, _ = self.comm.all_to_all_rows(back, counts)
        E.rows_scatter_add(asked, mine[:, :d], self.Ga)                 # Ga: scratch for Gstar rows
        E.rows_scatter_add(asked, mine[:, d:], self.Greg)
        E.div_scalar(self.Ga, float(self.L + 1), self.H)
        # --- backward hops: G_k = H + Aᵀ G_{k+1}
        g = self.H
        ping = (self.Ga, self.Gb)
        for k in range(self.L):
            self.comm.all_gather_rows(g, self.X)
            out = ping[(k + 1) & 1]
            self.At.matmul(self.X, out=out, addend=self.H)
            g = out
        E.adam_dense2(self.E0, self.m, self.v, g, self.Greg, self.adam)
        self.adam.advance()
        gs.zero_(); gr.zero_(); self.Greg.zero_(); self.Ga.zero_()    # buffers the head accumulates into
        return loss_out


class ShardedMF:
    """BPR-MF with both tables row-sharded (SURVEY §8e "BPR-MF step"): rank r owns the padded block
    [r·b, (r+1)·b) of the node rows (users first, then items) with its Adam moments.  One step
    (MF.py:54-76,101) = ids to the owners (all-to-all) -> the requested rows back (all-to-all) ->
    the BPR head on the compact [3B][d] block -> gradient rows to the owners (all-to-all),
    scatter-added (duplicates summed) -> owner-local TF-sparse Adam, which sweeps every local row
    (SURVEY H2).  Equals the single-process step on the concatenated global batch."""

    def __init__(self, comm, user_table, item_table, lr, reg, max_batch):
        dev = E.require_gpu()
        self.comm, self.rank, self.world = comm, comm.rank, comm.world
        ut = np.asarray(user_table, dtype=np.float32)
        it = np.asarray(item_table, dtype=np.float32)
        self.n_users, self.n_items, self.d = ut.shape[0], it.shape[0], ut.shape[1]
        self.N = self.n_users + self.n_items
        self.b = parallel.block_size(self.N, self.world)
        self.lo = min(self.rank * self.b, self.N)
        self.hi = min(self.lo + self.b, self.N)
        self.n_loc = self.hi - self.lo
        full = np.concatenate([ut, it])
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        self.T = z(self.b, self.d)                                   # my rows of [P ; Q]
        self.T[:self.n_loc] = torch.from_numpy(np.ascontiguousarray(full[self.lo:self.hi])).to(dev)
        self.m, self.v, self.G = z(self.b, self.d), z(self.b, self.d), z(self.b, self.d)
        self.reg, self.max_batch = float(reg), int(max_batch)
        self.adam = E.AdamState(lr)
        B3 = 3 * self.max_batch
        self.req = z(B3, self.d)
        self.gP, self.gQ = z(self.max_batch, self.d), z(2 * self.max_batch, self.d)
        self.terms = z(8 * self.max_batch)
        self._ar = torch.arange(2 * self.max_batch, dtype=torch.int32, device=dev)

    def step(self, users, pos, neg, loss_out):
        B, d = users.numel(), self.d
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        nodes = torch.cat([users.long(), pos.long() + self.n_users, neg.long() + self.n_users])
        owner = torch.div(nodes, self.b, rounding_mode="floor")
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=self.world)[:self.world].cpu().tolist()
        local = (nodes - owner * self.b)[order].to(torch.int32).contiguous()
        asked, asked_counts = self.comm.all_to_all_rows(local, counts)          # exchange 1: ids
        rows = torch.empty((asked.numel(), d), dtype=torch.float32, device=self.T.device)
        E.rows_gather(asked, self.T, rows)
        got, _ = self.comm.all_to_all_rows(rows, asked_counts)                  # exchange 2: rows
        req = self.req[:3 * B]
        req[order] = got
        # compact tables: P' = rows [0,B) (one per triplet), Q' = rows [B,3B) (pos then neg)
        P, Q = req[:B], req[B:3 * B]
        gP, gQ = self.gP[:B], self.gQ[:2 * B]
        E.bpr_mf_grad(P, Q, self._ar[:B], self._ar[:B], (self._ar[:B] + B).contiguous(), self.reg,
                      gP, gQ, self.terms, loss_out)
        back = torch.cat([gP, gQ])[order].contiguous()
        mine, _ = self.comm.all_to_all_rows(back, counts)                       # exchange 3: gradients
        E.rows_scatter_add(asked, mine, self.G)
        E.adam_sparse(self.T, self.m, self.v, self.G, self.adam)                # clears G
        self.adam.advance()
        gP.zero_(); gQ.zero_()

    def tables(self):
        """Full (P, Q) on every rank (one all-gather; evaluation entrance)."""
        full = torch.empty(self.b * self.world, self.d, dtype=torch.float32, device=self.T.device)
        self.comm.all_gather_rows(self.T, full)
        return full[:self.n_users], full[self.n_users:self.N]
