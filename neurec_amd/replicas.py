"""Mult-VAE over the GPUs of one node: data-parallel REPLICAS (last row — "replicas only").

MultiVAE.py:137-184 steps on batches of USER rows; a user's row is an independent unit and the model is eight small
dense variables (I x h twice, the rest tiny), so nothing is partitioned: every rank holds the whole model, encodes /
decodes ITS share of each global batch and the gradients are averaged with ONE all-reduce per step before the dense
TF-Adam update (MultiVAE.py:126-135: the loss is a MEAN over the batch, so the global gradient is the mean of the
ranks' gradients over equal shares; the L2 term is the same on every rank and survives the mean unchanged).  The
gradients of an engine are re-seated as views of one flat buffer: the exchange is a single collective over
2·I·h + O(h²) floats (gowalla, h = 32: 10.5 MB), no packing copies.

Equal to the single-process step on the concatenated batch up to the association of the batch sums (the single engine
adds B rows in one order, W replicas add B/W rows each and then the W partial sums): tests hold 1e-6 relative.  The
dropout mask is keyed by CSR position (the same on any rank); the sampling noise is keyed by the row's place in ITS
rank's batch — a replica run draws a different, equally distributed eps than the single process unless `eps_given`.
"""
import torch

from . import engine as E


class MultiVAEReplicas:
    """engine: trainer.MultiVAEEngine or vae_wide.MultiVAEWideEngine, built identically on every rank (same
    parameters, same train CSR)."""

    def __init__(self, comm, engine, decorrelate=False):
        """decorrelate: give every rank its own noise / dropout seed (the sampling noise is keyed by a row's place in its
        rank's batch: without this rank r's row j would draw rank 0's row j's eps)."""
        self.comm, self.eng = comm, engine
        if decorrelate and comm.rank:
            engine.seed = (int(engine.seed) + 0x9E3779B97F4A7C15 * comm.rank) & (2 ** 63 - 1)
            if hasattr(engine, "_step_args"):
                engine._step_args = None               # (the native step's argument block holds the seed)
        gs = engine.gradient_tensors()
        self.flat = torch.zeros(sum(g.numel() for g in gs), dtype=torch.float32, device=gs[0].device)
        views, off = [], 0
        for g in gs:
            views.append(self.flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        engine.set_gradient_tensors(views)
        self.exchange_bytes_per_step = self.flat.numel() * 4

    def share(self, rows):
        """this rank's contiguous share of a global batch of user rows (equal shares: the mean of the ranks'
        gradient means is the global mean only then)"""
        n, W = rows.numel(), self.comm.world
        if n % W:
            raise ValueError("a global batch of %d users does not split into %d equal shares" % (n, W))
        per = n // W
        return rows[self.comm.rank * per:(self.comm.rank + 1) * per]

    def step(self, rows, anneal, keep=0.8, drop_given=None, eps_given=None, want_loss=True):
        """One optimiser step of the GLOBAL batch `rows` (the same tensor on every rank; eps_given, if any, is the
        global batch's [B][z] noise)."""
        mine = self.share(rows)
        W, r = self.comm.world, self.comm.rank
        if eps_given is not None:
            per = rows.numel() // W
            eps_given = eps_given[r * per:(r + 1) * per].contiguous()
        self.eng.step(mine, anneal, keep=keep, drop_given=drop_given, eps_given=eps_given, want_loss=want_loss,
                      apply=False)
        if self.comm.live:
            self.comm.allreduce_sum_(self.flat)
            if W > 1:
                E.scale(self.flat, 1.0 / W, self.flat)
        self.eng.apply_gradients()

    def loss(self):
        """(neg-ELBO, neg_ll, KL) of the last global batch: the ranks' batch means averaged (host read; syncs)"""
        total, neg_ll, kl = self.eng.loss()
        t = torch.tensor([total, neg_ll, kl], dtype=torch.float64, device=self.flat.device)
        self.comm.allreduce_sum_(t)
        return tuple(float(x) / self.comm.world for x in t.cpu())
