"""Column-sharded LightGCN over the GPUs of one node: rank r holds columns [r·d/W, (r+1)·d/W) of the embedding
table, of its Adam moments and of every layer buffer — for ALL N nodes — and steps on the WHOLE global batch.

Why this partition (and not only the row partition of neurec_amd/sharded.py).  LightGCN's propagation
E^{k+1} = Â·E^k (LightGCN.py:132-149) is column-wise: column c of the product needs column c of the operand and
nothing else.  So are its backward hops, the gradient rows of the BPR head (g_b·(e_i − e_j) etc., LightGCN.py:156-166)
and TF's ApplyAdam.  With the columns cut across ranks every one of the 2·L sparse hops of a step runs WITHOUT any
exchange — where the row partition moves the whole [N][d] table through an all-gather per hop (5.6 GB at config 4;
xGMI is point-to-point at ≈50–60 GB/s per link and direction, so that all-gather costs more than the hop).  The
single quantity of a step that needs all d columns is the pair of inner products per triplet,
x_b = ⟨e_u, e_i⟩ − ⟨e_u, e_j⟩ (and the regulariser's sum of squares, for the logged loss): every rank forms its
partial products (12 B per triplet), ONE all-gather moves W·B·12 bytes (786 KB for a global batch of 8,192 on 8
ranks), every rank adds the W partials in rank order — the same additions in the same order everywhere, so x_b, the
loss and hence every rank's column slice of the update are consistent by construction — and the step continues
locally.  Nothing is computed twice: a rank gathers 1/W of the bytes of every row; only the CSR indices (8 B per
non-zero) are read by every rank.

What is exact.  Per column the arithmetic is the single-GPU engine's (same kernels on a narrower table; widths that
are not built run zero-padded, see trainer.LightGCNEngine).  The only difference to one GPU is the association of the
inner product: Σ over ranks of per-rank partial dots instead of one d-wide dot — last-ulp differences in x_b, well
inside north_star's 1e-5; at W = 1 the engine is bit-identical to LightGCNEngine (tests/test_colshard_gpu.py).

Every rank must be handed the SAME global batch (the sampler is a counter-based generator: same seed, same stream on
every rank — trainer.BprEpochSampler with rank=0, world=1 — so no ids travel either), with its batch plan.
"""
import numpy as np
import torch

from . import engine as E
from .trainer import LightGCNEngine


class ColumnShardedLightGCN:
    def __init__(self, comm, adj_csr, n_users, n_items, embed, n_layers, lr, reg, max_batch, adj_t_csr=None,
                 rank=None, world=None, keep_order=False):
        """embed: the full [N][d] table (each rank keeps its columns).  rank / world override comm's (a one-GPU run
        of ONE rank's share of a W-rank job: the measurement bench.py reports next to the single-GPU step)."""
        self.comm = comm
        self.rank = comm.rank if rank is None else int(rank)
        self.world = comm.world if world is None else int(world)
        emb = np.asarray(embed, dtype=np.float32) if not isinstance(embed, torch.Tensor) else embed
        self.d = int(emb.shape[1])
        if self.d % self.world:
            raise ValueError("embed_size %d is not a multiple of the %d ranks" % (self.d, self.world))
        self.d_loc = self.d // self.world
        lo = self.rank * self.d_loc
        self.local = LightGCNEngine(adj_csr, n_users, n_items, emb[:, lo:lo + self.d_loc], n_layers, lr, reg,
                                    max_batch, adj_t_csr=adj_t_csr, keep_order=keep_order)
        dev = self.local.E0.device
        self.max_batch = int(max_batch)
        self._parts = torch.zeros(3 * self.max_batch, dtype=torch.float32, device=dev)
        self._given = torch.zeros(3 * self.max_batch, dtype=torch.float32, device=dev)
        self.exchange_bytes_per_step = None

    @property
    def adam(self):
        return self.local.adam

    def step(self, users, pos, neg, loss_out=None, plan=None):
        """One optimiser step on the global batch (identical on every rank)."""
        B = users.numel()
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        ctx, n = self.local._ctx, 3 * B
        ctx.lightgcn_step_colshard_fwd(users, pos, neg, self._parts)
        if self.comm.live and self.world == self.comm.world:
            allp = torch.empty((self.world, n), dtype=torch.float32, device=self._parts.device)
            self.comm.all_gather_rows(self._parts[:n], allp)            # the step's ONE exchange: 12 B per triplet and rank
            E.partials_sum(allp, self.world, n, self._given)
            given = self._given
            self.exchange_bytes_per_step = self.world * n * 4
        else:
            given = self._parts                                          # one rank (or one rank's share): nothing to add
        ctx.lightgcn_step_colshard_bwd(users, pos, neg, self.local.adam, loss_out, plan, given)
        self.local.adam.advance()

    def final_embeddings(self):
        """Full (user, item) E* tables on every rank: one all-gather of the column slices (evaluation entrance)."""
        eu, ei = self.local.final_embeddings()
        if not (self.comm.live and self.world == self.comm.world):
            return eu, ei
        loc = torch.cat([eu, ei]).contiguous()                           # [N][d_loc]
        allc = torch.empty((self.world,) + tuple(loc.shape), dtype=loc.dtype, device=loc.device)
        self.comm.all_gather_rows(loc, allc.view(self.world * loc.shape[0], loc.shape[1]))
        full = allc.permute(1, 0, 2).reshape(loc.shape[0], self.d).contiguous()
        U = self.local.n_users
        return full[:U], full[U:]

    def table(self):
        """this rank's columns of E0 ([N][d_loc])"""
        return self.local.E0[:, :self.d_loc]
