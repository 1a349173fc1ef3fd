"""Device-resident training/evaluation engines for the hot path.

Host-side orchestration only: buffers live in HBM as torch tensors, every
arithmetic step is a HIP kernel launched through the C ABI (engine.py).  These
classes are what the drop-in model plugins (neurec_amd/model/general_recommender)
and bench.py drive.

  BprEpochSampler   PairwiseSampler.__iter__        data/sampler.py:198-206
  MFEngine          MF.build_graph + one sess.run   model/general_recommender/MF.py:45-76,101
  LightGCNEngine    LightGCN.build_graph + sess.run model/general_recommender/LightGCN.py:80-149,178
  FullRankEvaluator UniEvaluator.evaluate           evaluator/backend/cpp/uni_evaluator.py:101-157
"""
import os
import numpy as np
import torch

from . import engine as E


class TripletBatch(tuple):
    """(users, pos, neg) device views of one batch; `.plan` = the batch's sorted occurrence keys
    (engine.bpr_plan) or None; `.next_plan` = the plan of the batch that follows it in the epoch
    (None for the last).  Unpacks like the reference sampler's 3-tuples."""
    plan = None
    next_plan = None

    def __new__(cls, users, pos, neg, plan=None, next_plan=None):
        self = super().__new__(cls, (users, pos, neg))
        self.plan, self.next_plan = plan, next_plan
        return self


class BprEpochSampler:
    """Per-epoch BPR triplet stream generated on the device (no host round trip).  With
    `plan_users` = the number of user rows of the model that will consume the stream (neg_num = 1)
    every epoch also gets the batch plans of all its batches in one launch — the order in which
    the gradient kernels sum the occurrences of a row (engine.bpr_plan)."""

    def __init__(self, train_csr, n_items, neg_num=1, batch_size=1024, shuffle=True,
                 drop_last=False, seed=2018, rank=0, world=1, plan_users=None):
        if neg_num <= 0:
            raise ValueError("'neg_num' must be a positive integer.")
        if train_csr.nnz == 0:
            raise ValueError("'user_pos_dict' cannot be empty.")
        deg_max = int(np.diff(train_csr.h_indptr).max())
        if n_items <= deg_max:
            raise ValueError("The number of 'exclusion' is greater than 'high'.")
        self.csr, self.n_items, self.neg_num = train_csr, int(n_items), int(neg_num)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), drop_last
        self.seed, self.epoch = int(seed), 0
        self.row_of = train_csr.row_of()
        # multi-GPU: rank r owns the contiguous slice [lo, hi) of every epoch's stream
        n = train_csr.nnz
        self.lo = (n * rank) // world
        self.hi = (n * (rank + 1)) // world
        self.n_local = self.hi - self.lo
        dev = train_csr.indptr.device
        self._users = torch.empty(max(self.n_local, 1), dtype=torch.int32, device=dev)
        self._pos = torch.empty(max(self.n_local, 1), dtype=torch.int32, device=dev)
        self._neg = torch.empty(max(self.n_local * self.neg_num, 1), dtype=torch.int32, device=dev)
        self.plan_users = None if plan_users is None else int(plan_users)
        self.plans = self.plan_users is not None and self.neg_num == 1
        self._plan = (torch.empty(max(3 * self.n_local, 1), dtype=torch.int64, device=dev)
                      if self.plans else None)

    def __len__(self):
        if self.drop_last:
            return self.n_local // self.batch_size
        return (self.n_local + self.batch_size - 1) // self.batch_size

    def sample_epoch(self):
        """Fill the epoch buffers; returns (users, pos, neg) device tensors of the whole epoch."""
        out = E.sample_bpr_epoch(self.csr, self.row_of, self.n_items, self.neg_num, self.seed,
                                 self.epoch, self.shuffle, self.lo, self.n_local,
                                 out=(self._users, self._pos, self._neg))
        self.epoch += 1
        if self.plans and self.n_local > 0:
            E.bpr_plan(out[0], out[1], out[2], self.batch_size, self.plan_users, out=self._plan)
        return out

    def epoch_stream(self):
        """A fresh epoch as whole-stream device tensors (users, pos, neg, plans-or-None): what
        MFEngine.run_batches consumes in one native call."""
        users, pos, neg = self.sample_epoch()
        n = self.n_local
        return (users[:n], pos[:n], neg[:n * self.neg_num],
                self._plan[:3 * n] if self.plans else None)

    def batches(self):
        """Yield device-tensor batches (views, no copies) for one epoch."""
        users, pos, neg = self.sample_epoch()
        B = self.batch_size
        for k in range(len(self)):
            b, e = k * B, min((k + 1) * B, self.n_local)
            nb = neg[b * self.neg_num:e * self.neg_num]
            e2 = min((k + 2) * B, self.n_local)
            yield TripletBatch(users[b:e], pos[b:e],
                               nb if self.neg_num == 1 else nb.view(-1, self.neg_num),
                               self._plan[3 * b:3 * e] if self.plans else None,
                               self._plan[3 * e:3 * e2] if (self.plans and e2 > e) else None)


class MFEngine:
    """BPR-MF tables + TF-style Adam state in HBM; step() = one reference sess.run.

    lazy=True (default): TF-1.12's sparse Adam — which decays and moves EVERY row each step — is
    applied by exact lazy replay: a row's missed zero-gradient steps are replayed in registers when
    the row is next touched, bit-identical to sweeping the table every step (lazy=False, the
    checker).  lazy_period bounds how far a row may fall behind (rows r = t mod period are
    refreshed every step).  Two forms:
      fused=True (default): gradient AND optimiser in ONE launch (csrc/bpr.hip: mf_fused_step_kernel).
        Every table exists twice with a stamp per copy; readers of step t take the newer copy older
        than t, the one writer of a row (the head of its run in the batch plan, holding the summed
        gradient in registers) writes the other copy — no gradient table, no launch boundary.
      fused=False: nrhip_bpr_mf_grad_lazy then nrhip_adam_sparse_tf_lazy (two dependent launches).
    Measured on MI355X at the gowalla shape, B = 512, with the sampler's next-batch plans
    (scripts/exp_mf_fused.py): sweep 29.0-29.9 us/step; two launches: period 4: 20.2, 8: 19.0,
    16: 19.3, 32: 20.6; one launch: period 4: 14.6, 8: 12.8, 16: 13.0, 32: 15.8 us.  Without next
    plans the gradient waves redo the replay (a chain of exact fp32 sqrt + divide per missed step)
    for every row they gather (two launches: 22.7 / 27.1 / 48.4 us at period 4 / 16 / 64).
    The tables are only current after flush(); the P / Q / mP / ... properties flush for you."""

    # step-size table (4 MB).  Steps beyond it are served from its tail: TF's lr_t is ONE value once both
    # running fp32 powers vanish against 1 (t > ~17.3 k at beta2 = 0.999; checked below and again by the
    # native context), so a run of any length keeps stepping — conf/MF.properties' 300 epochs of gowalla
    # are 475 k steps, a larger dataset passes 2^20
    ALPHA_STEPS = 1 << 20

    def __init__(self, user_table, item_table, lr, reg, max_batch, lazy=True, lazy_period=None, fused=None):
        dev = E.require_gpu()
        ut = torch.as_tensor(user_table, dtype=torch.float32)
        it = torch.as_tensor(item_table, dtype=torch.float32)
        nu, rows = ut.shape[0], ut.shape[0] + it.shape[0]
        self.lazy = bool(lazy)
        # fused: gradient + lazy Adam in ONE launch on double-buffered tables (csrc/bpr.hip:
        # mf_fused_step_kernel); the two-launch lazy form stays for larger tables and as an A/B
        self.fused = self.lazy and (True if fused is None else bool(fused))
        # replay bound: measured best per form (12.5 us at 8 in one launch, 19.0 at 8-16 in two)
        self.lazy_period = int(lazy_period) if lazy_period is not None else (8 if self.fused else 16)
        # user and item tables (and their moments / gradients) share one allocation each, so the
        # TF-sparse Adam update of a step is a single launch over [U+I][d]
        if self.fused:
            self._w2 = torch.empty((2, rows, ut.shape[1]), dtype=torch.float32, device=dev)
            self._w2[0].copy_(torch.cat([ut, it]))
            self._w2[1].zero_()
            self._m2, self._v2 = torch.zeros_like(self._w2), torch.zeros_like(self._w2)
            self._table, self._m, self._v, self._g = self._w2[0], self._m2[0], self._v2[0], None
            self._tw = torch.tensor([0, -1], dtype=torch.int32, device=dev).repeat(rows, 1).contiguous()
            self._inb = torch.zeros(rows, dtype=torch.int32, device=dev)
            self.GP = self.GQ = None
        else:
            self._table = torch.cat([ut, it]).contiguous().to(dev)
            self._m, self._v, self._g = (torch.zeros_like(self._table) for _ in range(3))
            self.GP, self.GQ = self._g[:nu], self._g[nu:]
        self._P, self._Q = self._table[:nu], self._table[nu:]
        self._views = {"mP": self._m[:nu], "mQ": self._m[nu:], "vP": self._v[:nu], "vQ": self._v[nu:]}
        self.reg = float(reg)
        self.adam = E.AdamState(lr)
        self.terms = torch.empty(8 * max_batch, dtype=torch.float32, device=dev)
        self.max_batch = max_batch
        self._stale = False
        self._terms_steps = None
        self._alpha_host = None
        if self.lazy:
            self._alpha_np = self.adam.alpha_table(self.ALPHA_STEPS)
            self._alpha_tail_const = bool(self._alpha_np.size > 256 and
                                          np.all(self._alpha_np[-256:] == self._alpha_np[-1]))
            self._alpha_tab = torch.from_numpy(self._alpha_np).to(dev)
            if not self.fused:
                self._last = torch.zeros(rows, dtype=torch.int32, device=dev)
                self._stamp = torch.zeros(rows, dtype=torch.int32, device=dev)
        self._ctx = E.NativeStep.for_mf(self)

    # tables and moments as the sweep would have left them: brought up to date on access
    def _check_step_sizes(self, upto):
        if self.lazy and upto >= self._alpha_tab.numel() and not self._alpha_tail_const:
            raise NotImplementedError("step %d is beyond the step-size table (%d entries) and the table's tail "
                                      "is not constant yet: enlarge MFEngine.ALPHA_STEPS" % (upto, self.ALPHA_STEPS))

    def flush(self):
        if self.lazy and self._stale:
            self._ctx.mf_flush(self.adam)
            self._stale = False

    P = property(lambda self: (self.flush(), self._P)[1])
    Q = property(lambda self: (self.flush(), self._Q)[1])
    mP = property(lambda self: (self.flush(), self._views["mP"])[1])
    mQ = property(lambda self: (self.flush(), self._views["mQ"])[1])
    vP = property(lambda self: (self.flush(), self._views["vP"])[1])
    vQ = property(lambda self: (self.flush(), self._views["vQ"])[1])

    def step(self, users, pos, neg, loss_out, plan=None, next_plan=None):
        """One native call: fused gather/BPR/ordered row sums kernel + TF-sparse Adam (lazy replay on
        the touched and scheduled rows, or the full sweep).  loss_out: 2-float device tensor
        receiving (bpr_sum, reg_term); plan: the batch's TripletBatch.plan (None: sorted inside the
        step); next_plan: TripletBatch.next_plan — lazy mode brings the rows of the coming batch up to
        date in this step's optimiser launch, so its gradient kernel replays nothing (same results
        with or without)."""
        self._check_step_sizes(self.adam.t + 2)
        self._ctx.mf_step(users, pos, neg, self.adam, loss_out, plan, next_plan if self.lazy else None)
        self.adam.advance()
        self._stale = True

    def run_batches(self, users, pos, neg, batch, loss_steps, plans=None):
        """The batch loop of MF.train_model (MF.py:95-103) over the consecutive batches of an epoch
        stream (device tensors of the whole stream, `batch` triplets per step, the last batch short)
        in ONE native call — a Python loop enqueues ~12 us per step, more than the one-launch step
        takes.  loss_steps: float32 device tensor, 2 per step; plans: the stream's batch plans
        (BprEpochSampler: engine.bpr_plan over the whole stream) or None.  Returns the step count."""
        n = users.numel()
        n_steps = (n + batch - 1) // batch
        self._check_step_sizes(self.adam.t + n_steps + 1)
        if batch > self.max_batch:
            raise ValueError("batch larger than max_batch")
        if n_steps == 0:
            return 0
        t = self.adam.t
        if self.lazy:       # the device table's host twin; steps beyond it take its (constant) tail
            idx = np.minimum(np.arange(t + 1, t + 1 + n_steps), self._alpha_np.size - 1)
            h_alpha = np.ascontiguousarray(self._alpha_np[idx])
        else:
            if self._alpha_host is None or self._alpha_host.size < t + n_steps + 1:
                self._alpha_host = self.adam.alpha_table(max(2 * (t + n_steps), 4096))
            h_alpha = np.ascontiguousarray(self._alpha_host[t + 1:t + 1 + n_steps])
        # the fused steps leave their per-triplet terms for ONE reduction launch per native call: run the stream in
        # groups of kGroup steps so that buffer stays bounded (2 * batch * 4096 floats) whatever the epoch's length
        kGroup = 4096
        for s0 in range(0, n_steps, kGroup):
            ns = min(kGroup, n_steps - s0)
            lo, hi = s0 * batch, min((s0 + ns) * batch, n)
            terms = None
            if self.fused:
                if self._terms_steps is None or self._terms_steps.numel() < 2 * batch * ns:
                    self._terms_steps = torch.empty(2 * batch * ns, dtype=torch.float32, device=self._table.device)
                terms = self._terms_steps
            self._ctx.mf_steps(users[lo:hi], pos[lo:hi], neg[lo:hi], batch, self.adam,
                               np.ascontiguousarray(h_alpha[s0:s0 + ns]), loss_steps[2 * s0:2 * (s0 + ns)],
                               None if plans is None else plans[3 * lo:3 * hi], terms)
            for _ in range(ns):
                self.adam.advance()
        self._stale = True
        return n_steps

    def step_reference(self, users, pos, neg, loss_out, plan=None):
        """The same step as individual engine calls with the sweep (what nrhip_mf_step enqueues when
        lazy=False)."""
        if self.lazy:
            raise ValueError("step_reference sweeps the table: build the engine with lazy=False")
        if users.numel() > self.max_batch:
            raise ValueError("batch larger than max_batch")
        E.bpr_mf_grad(self._P, self._Q, users, pos, neg, self.reg, self.GP, self.GQ, self.terms,
                      loss_out, plan)
        E.adam_sparse(self._P, self._views["mP"], self._views["vP"], self.GP, self.adam)
        E.adam_sparse(self._Q, self._views["mQ"], self._views["vQ"], self.GQ, self.adam)
        self.adam.advance()


class GeneralMFEngine:
    """MF with every loss / optimiser combination conf/MF.properties allows (MF.py:62-76,
    util/learner.py): pairwise {bpr, hinge, square} on (user, pos, neg) triplets or pointwise
    {cross_entropy, square} on (user, item, label) instances; learner in {adam, gd, adagrad,
    rmsprop, momentum} with TF-1.12's *sparse* application (Adam sweeps every row, the others
    move only the rows the batch touched)."""

    def __init__(self, user_table, item_table, lr, reg, max_batch, loss="bpr", pairwise=True,
                 learner="adam", momentum=0.9):
        dev = E.require_gpu()
        loss, learner = str(loss).lower(), str(learner).lower()
        table = E.PAIRWISE_LOSSES if pairwise else E.POINTWISE_LOSSES
        if loss not in table:
            raise Exception("please choose a suitable loss function")        # learner.py:28,40
        if learner != "adam" and learner not in E.ROW_OPTIMIZERS:
            raise ValueError("please select a suitable optimizer")           # learner.py:15
        self.loss, self.pairwise, self.learner = loss, bool(pairwise), learner
        self.P = torch.as_tensor(user_table, dtype=torch.float32).contiguous().to(dev)
        self.Q = torch.as_tensor(item_table, dtype=torch.float32).contiguous().to(dev)
        self.GP, self.GQ = torch.zeros_like(self.P), torch.zeros_like(self.Q)
        self.reg, self.lr, self.momentum = float(reg), float(lr), float(momentum)
        self.adam = E.AdamState(lr)
        self.terms = torch.empty(8 * max_batch, dtype=torch.float32, device=dev)
        self.max_batch = max_batch
        self.flagP = torch.zeros(self.P.shape[0], dtype=torch.uint8, device=dev)
        self.flagQ = torch.zeros(self.Q.shape[0], dtype=torch.uint8, device=dev)
        init = {"adam": 0.0, "gd": None, "adagrad": 1e-8, "rmsprop": 1.0, "momentum": 0.0}[learner]
        mk = lambda t, v: None if v is None else torch.full_like(t, v)
        self.s0P, self.s0Q = mk(self.P, init), mk(self.Q, init)
        two = learner in ("adam", "rmsprop")
        self.s1P, self.s1Q = (mk(self.P, 0.0), mk(self.Q, 0.0)) if two else (None, None)

    def _apply(self, var, s0, s1, grad, flag):
        if self.learner == "adam":
            E.adam_sparse(var, s0, s1, grad, self.adam)
        elif self.learner == "rmsprop":
            E.optimizer_rows("rmsprop", var, s0, s1, grad, flag, self.lr, 0.9, 0.0, 1e-10)
        elif self.learner == "momentum":
            E.optimizer_rows("momentum", var, s0, None, grad, flag, self.lr, self.momentum)
        else:
            E.optimizer_rows(self.learner, var, s0, None, grad, flag, self.lr)

    def step(self, users, items, third, loss_out):
        """pairwise: third = negative items (int32); pointwise: third = labels (float32)."""
        if users.numel() > self.max_batch:
            raise ValueError("batch larger than max_batch")
        if self.pairwise:
            E.pairwise_mf_grad(self.P, self.Q, users, items, third, self.reg, self.loss, self.GP,
                               self.GQ, self.terms, loss_out)
        else:
            E.pointwise_mf_grad(self.P, self.Q, users, items, third, self.reg, self.loss, self.GP,
                                self.GQ, self.terms, loss_out)
        if self.learner != "adam":
            E.mark_rows(users, self.flagP)
            E.mark_rows(items, self.flagQ)
            if self.pairwise:
                E.mark_rows(third, self.flagQ)
        self._apply(self.P, self.s0P, self.s1P, self.GP, self.flagP)
        self._apply(self.Q, self.s0Q, self.s1Q, self.GQ, self.flagQ)
        self.adam.advance()


class LightGCNEngine:
    """LightGCN on one GPU: E0 (user rows then item rows), its Adam state, the
    normalised adjacency (and its transpose when not symmetric) and the layer
    buffers all stay resident in HBM."""

    def __init__(self, adj_csr, n_users, n_items, embed, n_layers, lr, reg, max_batch,
                 adj_t_csr=None, keep_order=False):
        """keep_order: the forward adjacency keeps its rows' storage order (graph.lightgcn_adjacency with
        tf_order=True: `gcmc` rows descending, as the reference hands them to TF)"""
        dev = E.require_gpu()
        self.n_users, self.n_items, self.n_layers = int(n_users), int(n_items), int(n_layers)
        self.N = self.n_users + self.n_items
        if adj_t_csr is None and not isinstance(adj_csr, E.SpmmCSR):
            from .graph import is_symmetric, transpose_csr
            if not is_symmetric(adj_csr):            # 'norm'/'gcmc'/'mean': backward needs A^T
                adj_t_csr = transpose_csr(adj_csr)
        self.A = adj_csr if isinstance(adj_csr, E.SpmmCSR) else \
            E.SpmmCSR.from_scipy(adj_csr, split_row=n_users, keep_order=keep_order)
        if adj_t_csr is None:
            self.At = self.A
        else:
            self.At = adj_t_csr if isinstance(adj_t_csr, E.SpmmCSR) else E.SpmmCSR.from_scipy(adj_t_csr, split_row=n_users)
        emb = torch.as_tensor(embed, dtype=torch.float32)
        assert emb.shape[0] == self.N
        # widths the kernels are built for; any other embed_size runs zero-padded to the next one: padded columns
        # stay exactly zero (the propagation is column-wise, a zero gradient leaves Adam's m = v = 0 and the update
        # 0 / (0 + eps) = 0) and add exact zeros to every dot product — the real columns see the same arithmetic
        self.d_real = int(emb.shape[1])
        fits = [w for w in (16, 32, 64, 128, 256) if w >= self.d_real]
        if not fits:
            raise NotImplementedError("LightGCN embed_size %d > 256 is not built" % self.d_real)
        self.d = fits[0]
        self.E0 = torch.zeros((self.N, self.d), dtype=torch.float32, device=dev)
        self.E0[:, :self.d_real] = emb.to(dev)
        z = lambda: torch.zeros_like(self.E0)
        self.m, self.v = z(), z()
        self.Ea, self.Eb = z(), z()          # ping-pong layer buffers
        self.Esum = z()                      # running sum over layers (E* = Esum/(L+1))
        self.Gstar, self.Greg = z(), z()     # dL/dE*, reg*E0 rows
        self.H, self.Ga, self.Gb = z(), z(), z()
        self.reg = float(reg)
        self.adam = E.AdamState(lr)
        self.terms = torch.empty(8 * max_batch, dtype=torch.float32, device=dev)
        self.max_batch = max_batch
        self.Esum_rows = z()                 # E-sum on the batch rows (training steps)
        self.batch_rows = torch.zeros(3 * max_batch, dtype=torch.int32, device=dev)
        self.row_flag = torch.zeros(self.N, dtype=torch.uint8, device=dev)
        self.Gsync = None                    # allocated on first multi-GPU step
        self._ctx = E.NativeStep.for_lightgcn(self)

    # -- forward: Esum = sum_k A^k E0  (LightGCN.py:132-149) -------------------------
    def propagate(self):
        if self.n_layers == 0:
            self.Esum.copy_(self.E0)
            return self.Esum
        src, acc_in = self.E0, self.E0
        bufs = (self.Ea, self.Eb)
        for k in range(self.n_layers):
            out = None if k == self.n_layers - 1 else bufs[k % 2]   # last layer: only the sum
            self.A.matmul(src, out=out, sum_in=acc_in, sum_out=self.Esum)
            src, acc_in = out, self.Esum
        return self.Esum

    def final_embeddings(self):
        """(user, item) E* tables = mean over layers — the assign_opt of LightGCN.py:110-116."""
        self.propagate()
        Estar = torch.empty_like(self.Esum)
        E.div_scalar(self.Esum, float(self.n_layers + 1), Estar)
        if self.d_real != self.d:
            Estar = Estar[:, :self.d_real].contiguous()
        return Estar[:self.n_users], Estar[self.n_users:]

    # -- one training step = sess.run(self.opt) (LightGCN.py:178) ---------------------
    def step(self, users, pos, neg, loss_out=None, grad_sync=None, plan=None):
        """One native call enqueues the whole step (csrc/step.hip); `step_reference` below is the
        same launch sequence spelled out in Python.  grad_sync(tensor): optional in-place
        all-reduce of dL/dE0 across ranks — the step is then cut at that one exchange point.
        plan: the batch's TripletBatch.plan (None: sorted inside the step, one more launch)."""
        if grad_sync is None:
            self._ctx.lightgcn_step(users, pos, neg, self.adam, loss_out, plan)
        else:
            if self.Gsync is None:
                self.Gsync = torch.zeros_like(self.E0)
            self._ctx.lightgcn_step_grad(users, pos, neg, loss_out, self.Gsync, plan)
            grad_sync(self.Gsync)                # summed over ranks (RCCL all-reduce)
            self._ctx.lightgcn_step_apply(self.Gsync, self.adam)
        self.adam.advance()

    def step_reference(self, users, pos, neg, loss_out=None, grad_sync=None, plan=None):
        """Same arithmetic as propagating everything, minus work whose result is never read or is
        known to be zero: the loss only reads E* on the 3B batch rows, so the LAST forward hop is
        formed for those rows only; dL/dE* is non-zero on those rows only, so the FIRST backward
        hop skips every all-zero source row.  With L=3 that is 4 full SpMM passes instead of 6."""
        B = users.numel()
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        L = self.n_layers
        rows = self.batch_rows[:3 * B]
        E.lightgcn_mark_batch(users, pos, neg, self.n_users, rows, self.row_flag)
        if L == 0:
            esum = self.E0
        else:
            src, acc_in = self.E0, self.E0
            bufs = (self.Ea, self.Eb)
            for k in range(L - 1):
                self.A.matmul(src, out=bufs[k % 2], sum_in=acc_in, sum_out=self.Esum)
                src, acc_in = bufs[k % 2], self.Esum
            if self.d >= 64:
                self.A.matmul(src, sum_in=acc_in, sum_out=self.Esum_rows, y_row_wanted=self.row_flag)
            else:
                self.A.matmul(src, sum_in=acc_in, sum_out=self.Esum_rows)
            esum = self.Esum_rows                 # valid on the batch rows, which is all that is read
        E.lightgcn_bpr_grad(esum, self.E0, self.n_users, L, users, pos, neg, self.reg,
                            self.Gstar, self.Greg, self.terms, loss_out, plan)
        # backward through mean + propagation: G_L = H, G_k = H + A^T G_{k+1}, H = Gstar/(L+1).
        # Gstar, Greg and H are non-zero on the batch rows only, so they are derived and
        # re-zeroed row-sparsely; the dense passes are the SpMM hops and Adam.
        E.rows_div(rows, self.Gstar, float(L + 1), self.H)
        g = self.H
        bufs = (self.Ga, self.Gb)
        for k in range(L):
            self.At.matmul(g, out=bufs[k % 2], addend=self.H,
                           x_row_nonzero=self.row_flag if (k == 0 and self.d >= 64) else None)
            g = bufs[k % 2]
        if grad_sync is not None and self.Gsync is None:
            self.Gsync = torch.zeros_like(self.E0)
        if grad_sync is None:
            E.adam_dense2(self.E0, self.m, self.v, g, self.Greg, self.adam)   # grad = g + Greg
        else:
            E.add(g, self.Greg, self.Gsync)      # total dL/dE0 of this rank
            grad_sync(self.Gsync)                # summed over ranks (RCCL all-reduce)
            E.adam_dense(self.E0, self.m, self.v, self.Gsync, self.adam, clear_grad=False)
        E.rows_clear(rows, self.d, (self.Gstar, self.Greg, self.H), self.row_flag)
        self.adam.advance()

    def step_bytes(self):
        """Algorithmic HBM bytes of one step AS IMPLEMENTED, every launch counted:
        L-1 full forward hops, the wanted-rows hop, the head's 9 row gathers + 2 row stores per
        triplet, the column-masked hop, L-2 full backward hops, and the last hop with the optimiser
        in it (CSR + operand read; E0, m, v read and written; the gradient never leaves registers).
        SURVEY §8d's figure for the reference's own graph (every hop full, Adam as a separate pass)
        is step_bytes_survey()."""
        L, nd4 = self.n_layers, self.N * self.d * 4
        full, masked = self.A.algorithmic_bytes(self.d), self.A.masked_bytes(self.d)
        if L == 0:
            return 7 * nd4
        head = self.max_batch * (11 * self.d * 4 + 12)
        fwd = (L - 1) * full + masked
        bwd = masked + max(L - 2, 0) * full if L >= 2 else 0
        last = self.A.nnz * 8 + (self.N + 1) * 4 + nd4 + 6 * nd4
        return fwd + head + bwd + last

    def step_bytes_survey(self):
        """2L full SpMM passes + dense Adam 7·N·d·4 + six [B][d] gathers."""
        nd4 = self.N * self.d * 4
        return 2 * self.n_layers * self.A.algorithmic_bytes(self.d) + 7 * nd4 + 6 * self.max_batch * self.d * 4


class FullRankEvaluator:
    """Full-rank evaluation on the device; only the per-user metric matrix (or its column sums)
    ever crosses PCIe.  Two designs with identical results (tests/test_eval_gpu.py):
      pruned=True  (default) tile maxima from the scoring loop -> the top_k+1 best 32-item tiles per
                   user rescored, ranked, measured; no [batch][I] score matrix (see _evaluate_pruned);
      pruned=False score GEMM -> -inf train mask -> top-K -> metrics on a materialised slab, batch
                   by batch (scoring of batch b+1 overlapped with the ranking of batch b)."""

    def __init__(self, train_csr, test_csr, metric_ids, top_k, batch_rows=2048, overlap=True,
                 pruned=True, strike_plan=True, search=None, extra_tiles=2):
        # how the pruned path FINDS the tiles it rescores: "int8" (the default; tables of <= 128 columns, wider ones take
        # "bf16") = the bounded filter on the int8 matrix cores (csrc/score_i8.hip: 15-bit fixed point, exact integer
        # accumulators, the bound derived from the quantisation), "bf16" = the same on the bf16 matrix cores
        # (csrc/score_bf16.hip, d <= 128) — either way every row is certified against its error bound or redone from
        # fp32 rows — and "fp32" = the fp32 MFMA loop (exact maxima).  The ranked scores are the fp32 chain's in all
        # three.  NEUREC_EVAL_SEARCH overrides the default.
        self.search = str(search or os.environ.get("NEUREC_EVAL_SEARCH", "int8"))
        if self.search not in ("bf16", "int8", "fp32"):
            raise ValueError("search must be 'bf16', 'int8' or 'fp32', got %r" % (self.search,))
        self.extra_tiles = int(extra_tiles)  # bounded search: tiles rescored beyond top_k + 1 (room for the bound)
        self._filter, self._filters = None, {}
        # the int8 bound is 2-4 x the bf16 form's: where scores crowd together (an untrained NGCF: a third of the rows)
        # its certificate fails row after row and every such row costs a full fp32 row.  An evaluation that had to
        # redo rows under int8 sends the next `int8_retry` evaluations through bf16 before int8 is tried again.
        self.int8_retry = 8
        self._int8_pause = 0
        # ... and the int8 certificate gets more rescored tiles to stand on: with 23 tiles 1-3 of 29,858 rows of plain
        # gaussian tables stay uncertified (each costs a full fp32 row: 0.87 instead of 0.67 ms), with 25 none; four
        # more tiles cost 0.01 ms (profiles/r05_exp_filter_i8.txt)
        self.int8_extra_tiles = 4
        self.native_loop = True              # nrhip_eval_pruned (False, tests: the same entry points issued batch by batch)
        self._native_sums = None
        self.pruned = bool(pruned)           # tile-pruned path: no score matrix (see _evaluate_pruned)
        self.strike_plan = bool(strike_plan)  # strikes as a planned fix-up pass after an unmasked scoring loop
        self._plan = None                    # (False: cursors + strikes inside the scoring loop; same M bit for bit)
        self.train, self.test = train_csr, test_csr
        self.metric_ids = [int(m) for m in metric_ids]
        self.top_k = int(top_k)
        self.batch_rows = int(batch_rows)
        self.overlap = bool(overlap)         # score batch b+1 (MFMA / HBM-write bound) while batch b
        self._gemm = None                    # is being ranked (HBM-read bound) on a second stream
        self._scores = None
        self._side = None
        self._flags = None                   # pruned path: per-user "ranking may depend on ties" flags of the last run

    def evaluate_factors(self, user_table, item_table, test_users, exact_mean=False, per_user=False, column_sums=False):
        """Returns float64 column means [n_metric*top_k] (or the fp32 np.mean when
        exact_mean) over `test_users` (int32 device tensor); per_user: the [n][n_metric*top_k] fp32
        matrix cpp_evaluate_matrix returns (evaluate.h:53-72), one row per test user; column_sums: the fp64 column
        SUMS, undivided (a rank's share of a sharded evaluation: the ranks' sums are added, then divided once)."""
        want_rows = per_user
        n = test_users.numel()
        nm = len(self.metric_ids)
        if self._gemm is None or self._gemm.cols != item_table.shape[0] or \
                self._gemm.d != item_table.shape[1]:
            self._gemm = E.score_gemm_for(item_table, self.batch_rows)       # (loads the item side itself)
            # score slabs [slab_rows][I]: the materialised path's batch, the pruned path's redo rows — made when first
            # needed, and never wider than 1 GiB (at 10^6 items a batch_rows slab would be 32 GB for rows that are
            # almost never redone)
            self._scores = []
            self._filter = None
            self._gemm_stale = False
        else:
            # the scoring engine still holds the previous table's item copies: reloaded by whoever scores first — the
            # native pruned evaluation does it inside its one call (r06: a Python-level launch here, another for the
            # filter, then the native call left the device idle between them: 0.71 ms where the launches take 0.64)
            self._gemm_stale = True
        if want_rows or exact_mean or getattr(self, "_rows_buf", None) is None or \
                self._rows_buf.shape != (n, nm * self.top_k) or self._rows_buf.device != test_users.device:
            per_user = torch.empty((n, nm * self.top_k), dtype=torch.float32, device=test_users.device)
            if not (want_rows or exact_mean):
                self._rows_buf = per_user                 # (the sums-only form: the rows never leave this object)
        else:
            per_user = self._rows_buf
        self._flags, self.n_flagged, self.n_uncertified, self._native_sums = None, 0, 0, None
        self._n_rows, self._search_macs = n, float(n) * item_table.shape[0] * item_table.shape[1]
        cols = item_table.shape[0]
        starts = list(range(0, n, self.batch_rows))
        if self.pruned and 2 * ((cols + 63) // 64) >= self.top_k + 2 and self.top_k <= 62 and n > 0 and \
                not getattr(self._gemm, "wide", False):
            self._evaluate_pruned(user_table, item_table, test_users, per_user, starts)
        elif not self.overlap or len(starts) < 2:
            self._reload_items(item_table)
            for b in starts:
                u = test_users[b:b + self.batch_rows]
                S = self._gemm(user_table, u, out=self._slab(0, self.batch_rows))
                E.mask_train(S, u, self.train, cols=cols)
                E.eval_scores(S, self.test, self.metric_ids, self.top_k, users=u, cols=cols,
                              out=per_user[b:b + u.numel()])
        else:
            # two score slabs, two HIP streams: GEMM + mask of batch b+1 on the current stream,
            # top-K + metrics of batch b on the side stream; events order the slab hand-offs
            self._reload_items(item_table)
            self._slab(0, self.batch_rows)
            self._slab(1, self.batch_rows)
            if self._side is None:
                self._side = torch.cuda.Stream(device=test_users.device)
            main, side = torch.cuda.current_stream(), self._side
            ranked = [None, None]                       # per slab: event "ranking finished"
            for k, b in enumerate(starts):
                u = test_users[b:b + self.batch_rows]
                slab = self._scores[k % 2]
                if ranked[k % 2] is not None:
                    main.wait_event(ranked[k % 2])      # the slab is free again
                S = self._gemm(user_table, u, out=slab)
                E.mask_train(S, u, self.train, cols=cols)
                scored = main.record_event()
                with torch.cuda.stream(side):
                    side.wait_event(scored)
                    E.eval_scores(S, self.test, self.metric_ids, self.top_k, users=u, cols=cols,
                                  out=per_user[b:b + u.numel()])
                    ranked[k % 2] = side.record_event()
            main.wait_stream(side)
        if exact_mean or want_rows:
            if self._native_sums is not None:             # the counts lie behind the sums: the redo needs no host sync of its own
                both = self._read_native_sums()
                self.n_flagged, self.n_uncertified = int(both[-2]), int(both[-1])
                if self.n_flagged:
                    self._redo_native(item_table)
                self._flags = None
            else:
                self._redo_flagged(user_table, item_table, test_users, per_user)
            self._note_flags()
            rows = per_user.cpu().numpy()
            return rows if want_rows else np.mean(rows, axis=0)   # uni_evaluator.py:150-151
        div = 1 if column_sums else n
        # ONE device->host copy per evaluation: the column sums and the number of rows flagged for ties
        # travel together; only if some row was flagged are those rows redone and the sums retaken
        if self._native_sums is not None:                 # nrhip_eval_pruned left the sums and the flag count together
            both = self._read_native_sums()
            self.n_flagged, self.n_uncertified = int(both[-2]), int(both[-1])
            self._note_flags()
            if self.n_flagged:
                # the flagged rows again from full score rows and the sums retaken, in one native call (r06: the
                # Python-level redo — index, GEMM, mask, rank, scatter, sums — cost 0.31 ms whatever the count)
                self._redo_native(item_table)
                both = self._read_native_sums()
                self._flags = None
            return both[:-2] / div
        else:
            sums = E.colsum(per_user)
            if self._flags is None:
                return sums.cpu().numpy() / div    # every mean is the fp64 column sum divided ON THE HOST (a device-side
                #                                    scalar division may be a reciprocal multiply: last-ulp differences)
            both = torch.cat([sums.reshape(-1), (self._flags != 0).sum().to(sums.dtype).reshape(1),
                              ((self._flags & 2) != 0).sum().to(sums.dtype).reshape(1)]).cpu().numpy()
        self.n_flagged, self.n_uncertified = int(both[-2]), int(both[-1])
        self._note_flags()
        if self.n_flagged:
            # (both counts are known: the redo below runs without a host round trip of its own)
            self._redo_flagged(user_table, item_table, test_users, per_user, known=(self.n_flagged, self.n_uncertified))
            return E.colsum(per_user).cpu().numpy() / div
        return both[:-2] / div

    def _read_native_sums(self):
        """[column sums | flagged rows | rows among them whose certificate failed] of the native evaluation, on the host"""
        host = getattr(self, "_sums_host", None)
        if host is None or host.shape != self._native_sums.shape:
            host = self._sums_host = torch.empty(self._native_sums.shape, dtype=self._native_sums.dtype, pin_memory=True)
        host.copy_(self._native_sums, non_blocking=True)         # (pinned: no staging copy behind the device's)
        torch.cuda.current_stream().synchronize()
        self._native_sums = None
        return host.numpy().copy()

    def _redo_native(self, item_table):
        """nrhip_eval_redo: the rows the native evaluation flagged, from full fp32 score rows (at most 1 GiB of them at a
        time), over their rows of its output; leaves the retaken sums for _read_native_sums"""
        step = max(1, min(self.batch_rows, (1 << 28) // max(self._gemm.ld, 1)))
        slab = self._slab(0, min(step, self.n_flagged))
        stale = getattr(self, "_gemm_stale", False)
        self._native_sums = self._native.redo(self.n_flagged, slab, reload_items=2 if stale == "operand copy" else 1 if stale else 0)
        self._gemm_stale = False

    def _reload_items(self, item_table):
        """the scoring engine's item copies, if they are still the previous table's"""
        if getattr(self, "_gemm_stale", False):
            self._gemm.prepare(item_table)
            self._gemm_stale = False

    def _slab(self, k, rows):
        """score slab k with room for `rows` rows"""
        while len(self._scores) <= k:
            self._scores.append(None)
        if self._scores[k] is None or self._scores[k].shape[0] < rows:
            self._scores[k] = None
            self._scores[k] = self._gemm.new_score_buffer(rows)
        return self._scores[k]

    def _note_flags(self):
        # only rows whose CERTIFICATE failed (flag bit 1) say something about the int8 bound; rows flagged for ties or a
        # full bucket would be flagged under any search (a dataset with tied rows must not lose int8 for good)
        # ... and what a redone row costs is a fixed ~0.2 ms (a host round trip, three launches) plus one fp32 row: at the
        # gowalla shape that is more than int8 saves over bf16 (0.1 ms), at a search of tens of milliseconds (config 4:
        # 8,192 users x 10^6 items x 128) it is nothing — there only a run that fails every hundredth row pauses int8
        if getattr(self, "search_used", None) == "int8" and self.search == "int8":
            big = self._search_macs > 4e12
            if self.n_uncertified > (self._n_rows // 100 if big else 0):
                self._int8_pause = self.int8_retry

    def _evaluate_pruned(self, user_table, item_table, test_users, per_user, starts):
        """Level 1: tile maxima from the scoring loop (scores never stored); level 2: rescore and
        rank the top_k+1 best tiles per user.  Rows whose ranking could depend on ties come back
        flagged and are recomputed from full score rows — same numbers as the materialised path."""
        n = test_users.numel()
        # (every row's flag is WRITTEN by level 2 — remap_rank_kernel — so no fill: in steady state an evaluation
        #  launches nothing but this package's kernels; `torch.unique` / the user -> row table below run once per user list)
        flags = getattr(self, "_flags_buf", None)             # (read back before the next evaluation writes it)
        if flags is None or flags.numel() != n or flags.device != test_users.device:
            flags = self._flags_buf = torch.empty(n, dtype=torch.int32, device=test_users.device)
        self.n_flagged = 0
        plan = row_of = None
        use_plan = self.strike_plan
        if use_plan:
            # the planned strikes look a user's row up: test_users must be distinct (uni_evaluator.py:108 hands over
            # the keys of a dict).  Checked once per user list (one count, one host read), not per evaluation; a list
            # with repeats takes the in-loop strikes, which have no such precondition.
            # (keyed by the tensor object — held, so its address cannot be reused — and its in-place version counter)
            tag = (id(test_users), test_users._version, n)
            if getattr(self, "_distinct_tag", None) != tag:
                self._distinct_tag, self._users_ref = tag, test_users
                self._distinct = int(torch.unique(test_users).numel()) == n
            use_plan = self._distinct
        if use_plan:
            if self._plan is None or self._plan.cols != item_table.shape[0]:
                self._plan = E.TileStrikePlan(self.train, item_table.shape[0])
            plan = self._plan
            if getattr(self, "_row_of_tag", None) != tag:     # the user -> evaluation row table, once per user list
                t = torch.full((self.train.n_rows,), -1, dtype=torch.int32, device=test_users.device)
                t[test_users.long()] = torch.arange(n, dtype=torch.int32, device=test_users.device)
                self._row_of_tag, self._row_of = tag, t
            row_of = self._row_of
        filt = None
        arith = self.search
        if arith == "int8" and not E.ScoreFilter.supports(item_table.shape[1], "int8"):
            arith = "bf16"                                    # (both forms are built for d <= 128)
        if arith == "int8" and self._int8_pause > 0:
            self._int8_pause -= 1
            arith = "bf16"                                    # the last int8 evaluation left rows uncertified
        # the int8 bound is wider: its certificate gets more rescored tiles to stand on (int8_extra_tiles more)
        extra = self.extra_tiles + (self.int8_extra_tiles if arith == "int8" else 0)
        n_keep = min(self.top_k + 1 + extra, 63, 2 * ((item_table.shape[0] + 63) // 64) - 1)
        if arith != "fp32" and use_plan and E.ScoreFilter.supports(item_table.shape[1], arith) and n_keep > self.top_k:
            if self._filter is None:
                self._filters = {}                            # (a new scoring engine: new table shape)
            filt_stale = arith in self._filters               # both forms stay built: a paused int8 comes back
            if not filt_stale:
                self._filters[arith] = E.ScoreFilter(item_table, self.batch_rows, arith)     # (loads the item side itself)
            filt = self._filter = self._filters[arith]
        else:
            filt_stale = False
        self.search_used = filt.arith if filt is not None else "fp32"
        # the item side of both engines inside the native call when both are stale (the steady state); else from here
        native = use_plan and self.native_loop
        inside = native and self._gemm_stale and (filt is None or filt_stale)
        if not inside:
            self._reload_items(item_table)
            if filt is not None and filt_stale:
                filt.prepare(item_table)
        if use_plan and self.native_loop:
            # the whole batch loop, the column sums and the flagged-row count in one native call (nrhip_eval_pruned)
            keep = n_keep if filt is not None else self.top_k + 1
            key = (id(self._gemm), id(filt), id(plan), keep)
            if getattr(self, "_native_key", None) != key:
                cache = getattr(self, "_natives", None)
                if cache is None or cache[0] is not self._gemm:
                    cache = self._natives = (self._gemm, {})
                if key not in cache[1]:
                    cache[1][key] = E.PrunedEvaluation(self._gemm, filt, plan, self.train, self.test, self.metric_ids,
                                                       self.top_k, keep, self.batch_rows)
                self._native, self._native_key = cache[1][key], key
            # (with a filter the call leaves the fp32 scoring loop's operand copy alone — 2 — and the engine stays
            #  stale: rows redone from full score rows reload it first, _reload_items)
            _, _, self._native_sums = self._native.run(user_table, item_table, test_users, row_of, per_user, flags,
                                                       prepare_items=(2 if filt is not None else 1) if inside else 0)
            if not (inside and filt is not None):
                self._gemm_stale = False
            else:
                self._gemm_stale = "operand copy"             # (the k-major copy is this table's)
            self._flags = flags
            return
        self._native_sums = None
        for b in starts:
            u = test_users[b:b + self.batch_rows]
            if filt is None:
                M = self._gemm.tile_maxima(user_table, u, self.train, plan=plan, row_of=row_of, row_lo=b)
                E.eval_tiles(M, user_table, self._gemm, u, self.train, self.test, self.metric_ids,
                             self.top_k, per_user[b:b + u.numel()], flags[b:b + u.numel()])
            else:
                M, eps = self._gemm.tile_maxima(user_table, u, self.train, plan=plan, row_of=row_of, row_lo=b,
                                                filt=filt)
                E.eval_tiles(M, user_table, self._gemm, u, self.train, self.test, self.metric_ids,
                             self.top_k, per_user[b:b + u.numel()], flags[b:b + u.numel()], eps=eps, n_keep=n_keep)
        self._flags = flags                        # read by evaluate_factors together with the sums

    def _redo_flagged(self, user_table, item_table, test_users, per_user, known=None):
        """Rows whose ranking could depend on ties: recomputed from full score rows.  known = (flagged rows, rows among
        them whose certificate failed) when the caller has read the counts already (they come back with the column
        sums): the row list is then formed on the device without a host round trip."""
        if self._flags is None:
            return
        self._reload_items(item_table)
        cols = item_table.shape[0]
        redo = None
        if known is not None and known[0] > 0:
            try:
                redo = torch.nonzero_static(self._flags, size=int(known[0])).flatten()
                self.n_uncertified = int(known[1])
            except (RuntimeError, NotImplementedError):      # (a build without the static form on this device)
                redo = None
        if redo is None:
            redo = torch.nonzero(self._flags, as_tuple=False).flatten()      # host sync: only when rows were flagged
            self.n_uncertified = int(((self._flags & 2) != 0).sum()) if redo.numel() else 0
        self._flags = None
        self.n_flagged = int(redo.numel())
        if self.n_flagged:
            fixed = torch.empty((self.n_flagged, per_user.shape[1]), dtype=torch.float32,
                                device=per_user.device)
            # full fp32 rows, at most 1 GiB of them at a time
            step = max(1, min(self.batch_rows, (1 << 28) // max(self._gemm.ld, 1)))
            for lo in range(0, self.n_flagged, step):
                idx = redo[lo:lo + step]
                u = test_users[idx].contiguous()
                S = self._gemm(user_table, u, out=self._slab(0, min(step, self.n_flagged)))
                E.mask_train(S, u, self.train, cols=cols)
                E.eval_scores(S, self.test, self.metric_ids, self.top_k, users=u, cols=cols,
                              out=fixed[lo:lo + u.numel()])
            per_user[redo] = fixed                                       # plumbing copy of the rows


class NGCFEngine:
    """NGCF (alg_type=ngcf) on one GPU: ego embeddings E0 [N][d], per-layer weights
    (W_gc, b_gc, W_bi, b_bi), the (row-normalised, non-symmetric) adjacency and its transpose.
    A step = L sparse hops + L fused dense layers forward, the BPR head on the concatenated
    output, the same backwards, dense TF-Adam on E0 and on every weight
    (model/general_recommender/NGCF.py:91-110,160-202)."""

    def __init__(self, adj, adj_t, n_users, n_items, embed, weights, lr, reg, mess_dropout,
                 max_batch, seed=2017, learner="adam"):
        dev = E.require_gpu()
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.N = self.n_users + self.n_items
        self.A = E.SpmmCSR.from_scipy(adj, split_row=n_users)
        self.At = E.SpmmCSR.from_scipy(adj_t, split_row=n_users)
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)
        self.E0 = f(embed)
        self.d = self.E0.shape[1]
        self.L = len(weights)
        self.W = [tuple(f(np.reshape(w, -1) if w.ndim == 2 and w.shape[0] == 1 else w) for w in ws)
                  for ws in weights]
        for ws in self.W:
            assert ws[0].shape == (self.d, self.d), "layer width must equal the embedding size (16)"
        self.dsum = self.d * (self.L + 1)
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        self.Out, self.dOut = z(self.N, self.dsum), z(self.N, self.dsum)
        self.S = [z(self.N, self.d) for _ in range(self.L)]
        self.ego = [self.E0] + [z(self.N, self.d) for _ in range(self.L)]
        self.mask = [torch.zeros(self.N, self.d, dtype=torch.uint8, device=dev) for _ in range(self.L)]
        self.dS, self.dEd, self.dT1, self.dT2 = (z(self.N, self.d) for _ in range(4))
        self.dEgo = [z(self.N, self.d), z(self.N, self.d)]
        self.gE0 = z(self.N, self.d)
        self.mE, self.vE = z(self.N, self.d), z(self.N, self.d)
        self.gW = [tuple(torch.zeros_like(w) for w in ws) for ws in self.W]
        self.mW = [tuple(torch.zeros_like(w) for w in ws) for ws in self.W]
        self.vW = [tuple(torch.zeros_like(w) for w in ws) for ws in self.W]
        self.keep = 1.0 - float(mess_dropout)
        self.reg, self.seed, self.t = float(reg), int(seed), 0
        self.adam = E.AdamState(lr)
        # learner.py:2-17: adam runs on the engine's own kernels (and inside the native step); the other four are
        # applied tensor by tensor after the gradients of the spelled-out step (their slots live in the moment buffers)
        self.learner = E.make_learner(learner, lr)
        if self.learner is not None:
            self.learner.init_slots([self.mE] + [m for ms in self.mW for m in ms], [self.vE] + [v for vs in self.vW for v in vs])
        self.terms = torch.empty(8 * max_batch, dtype=torch.float32, device=dev)
        self.rows = torch.zeros(3 * max_batch, dtype=torch.int32, device=dev)
        self.flag = torch.zeros(self.N, dtype=torch.uint8, device=dev)
        self.ws = E.ngcf_workspace(self.N, dev)
        self.max_batch = max_batch
        # the ~27 launches of a step go out in ONE native call (a Python loop issues them in ~270 us,
        # more than they take on the GPU); contexts exist for at most E.NGCF_MAX_LAYERS layers
        self._ctx = E.NativeStep.for_ngcf(self) if (self.L <= E.NGCF_MAX_LAYERS and self.learner is None) else None

    def forward(self, masks=None):
        """Fills self.Out = concat(E0, out_1..out_L).  masks: optional list of uint8 [N][d] device
        tensors (tests); otherwise a fresh dropout draw per call — evaluation included, as in the
        reference (NGCF.py:193 has no training flag)."""
        d = self.d
        if self._ctx is not None:
            if masks is not None:
                for k in range(self.L):
                    self.mask[k].copy_(masks[k])
            self._ctx.ngcf_forward(self.seed, self.t, masks is not None)
            self.t += 1
            return self.Out
        E.copy2d(self.E0, self.Out[:, :d])
        for k in range(self.L):
            self.A.matmul(self.ego[k], out=self.S[k])
            if masks is not None:
                self.mask[k].copy_(masks[k])
            E.ngcf_layer_fwd(self.ego[k], self.S[k], self.W[k], self.keep, self.mask[k],
                             masks is not None, self.seed, self.t, k, self.ego[k + 1],
                             self.Out[:, (k + 1) * d:(k + 2) * d])
        self.t += 1
        return self.Out

    def final_embeddings(self):
        out = self.forward()
        return out[:self.n_users], out[self.n_users:]

    def step(self, users, pos, neg, loss_out, masks=None, plan=None):
        B, d = users.numel(), self.d
        if B > self.max_batch:
            raise ValueError("batch larger than max_batch")
        if self._ctx is not None:
            if masks is not None:
                for k in range(self.L):
                    self.mask[k].copy_(masks[k])
            self._ctx.ngcf_step(users, pos, neg, self.adam, self.seed, self.t, masks is not None, loss_out, plan)
            self.t += 1
            self.adam.advance()
            return
        self.forward(masks)
        U = self.n_users
        rows = self.rows[:3 * B]
        E.lightgcn_mark_batch(users, pos, neg, U, rows, self.flag)
        # the BPR head of NGCF.py:91-100 is the MF head on the rows of the concatenated output
        E.bpr_mf_grad(self.Out[:U], self.Out[U:], users, pos, neg, self.reg, self.dOut[:U],
                      self.dOut[U:], self.terms, loss_out, plan)
        dego = None
        for k in range(self.L - 1, -1, -1):
            E.ngcf_layer_bwd(self.ego[k], self.S[k], self.W[k], self.keep, self.mask[k],
                             self.dOut[:, (k + 1) * d:(k + 2) * d], dego, self.dS, self.dEd,
                             self.dT1, self.dT2, self.gW[k], self.ws)
            nxt = self.dEgo[k % 2]
            self.At.matmul(self.dS, out=nxt, addend=self.dEd)     # dE_k = dBi⊙S + A^T dS
            dego = nxt
        if dego is None:
            E.copy2d(self.dOut[:, :d], self.gE0)
        else:
            E.add2d(self.dOut[:, :d], dego, self.gE0)
        # every trainable in one launch (17 tensors = 2 launches; they were 17)
        tensors = [(self.E0, self.mE, self.vE, self.gE0)] + \
            [(w, m, v, g) for k in range(self.L) for w, m, v, g in zip(self.W[k], self.mW[k], self.vW[k], self.gW[k])]
        if self.learner is None:
            E.adam_dense_multi(tensors, self.adam)
        else:
            self.learner.apply(tensors)
        E.rows_clear(rows, self.dsum, (self.dOut,), self.flag)
        self.adam.advance()


class MultiVAEEngine:
    """Mult-VAE (model/general_recommender/MultiVAE.py) for the two-layer shape of
    conf/MultiVAE.properties: I -> h -> [mu|logvar] (2z) and z -> h -> I.

    Parameters live on the GPU as W_q0 [I][h], b_q0 [h], W_q1 [h][2z], b_q1 [2z], W_p0 [z][h],
    b_p0 [h], W_p1ᵀ [I][h] (item-major — the transpose of the TF variable), b_p1 [I].
    A step = encode (bag-sum over each user's CSR row; no dense [B][I] input) -> the decoder's loss and
    gradients with the logits recomputed tile by tile on the matrix cores, never stored (csrc/vae_fused.hip;
    `decoder="slab"` keeps the first form: one [B][I] logits slab written once, read three times) -> the
    narrow layers -> dense TF-Adam on all eight variables (MultiVAE.py:126-139)."""

    NAMES = ("Wq0", "bq0", "Wq1", "bq1", "Wp0", "bp0", "Wp1t", "bp1")

    def __init__(self, train_csr, n_items, params, lr, reg, act, max_batch, seed=2017, decoder=None, learner="adam"):
        dev = E.require_gpu()
        if decoder is None:                      # A/B switch for tests and profiles; the product default is "fused"
            import os
            decoder = os.environ.get("NEUREC_VAE_DECODER", "fused")
        if decoder not in ("fused", "slab"):
            raise ValueError("decoder must be 'fused' or 'slab'")
        self.csr, self.n_items = train_csr, int(n_items)
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)
        self.P = {k: f(params[k]) for k in self.NAMES}
        self.h, self.z = self.P["Wp0"].shape[1], self.P["Wp0"].shape[0]
        if self.h > 32 or 2 * self.z > 32:
            # every kernel of this engine (csrc/vae.hip, vae_fused.hip) is written for hidden <= 32 and latent <= 16;
            # said here, not by an opaque NR_ERR_UNSUPPORTED at the first step
            raise ValueError("MultiVAEEngine: hidden width %d / latent %d outside the narrow engine's range (h <= 32, "
                             "z <= 16): use vae_wide.MultiVAEWideEngine, as the MultiVAE plugin does" % (self.h, self.z))
        self.decoder = decoder
        assert self.P["Wq0"].shape == (self.n_items, self.h)
        assert self.P["Wq1"].shape == (self.h, 2 * self.z)
        assert self.P["Wp1t"].shape == (self.n_items, self.h)
        self.G = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.M = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.V = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.act, self.reg, self.seed, self.t = act, float(reg), int(seed), 0
        self.adam = E.AdamState(lr)
        self.max_batch = int(max_batch)
        B, h, z = self.max_batch, self.h, self.z
        zf = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.H1, self.G1, self.dG1, self.DA3, self.DA1 = (zf(B, h) for _ in range(5))
        self.MU, self.LOGVAR, self.EPSSTD, self.ZS = (zf(B, z) for _ in range(4))
        self.DH2 = zf(B, 2 * z)
        self.KLb, self.nll = zf(B), zf(B)
        self.h0val = zf(max(train_csr.nnz, 1))
        self.gemm = E.ScoreGemm(self.P["Wp1t"], B)
        self._S = None                             # the [B][I] slab exists only where somebody asks for logits
        self.ws = (E.vae_fused_workspace(B, self.n_items, dev) if decoder == "fused"
                   else E.vae_workspace(B, self.n_items, dev))
        self.stats = zf(2)                         # [neg_ll, KL] of the last step
        self.regsum = torch.zeros(1, dtype=torch.float64, device=dev)
        # the whole step as one native call (nrhip_vae_step; False, tests: the same entry points issued from Python)
        self.native_step = decoder == "fused"
        self._step_args, self._step_key = None, None
        self.learner = E.make_learner(learner, lr)          # learner.py:2-17; None: adam (the engine's own kernels)
        if self.learner is not None:
            self.learner.init_slots(self.M.values(), self.V.values())

    @property
    def S(self):
        if self._S is None:
            self._S = self.gemm.new_score_buffer(self.max_batch)
        return self._S

    def gemm_out(self, rows):
        """a fresh [rows][ld] score slab (predict hands its result to the evaluator)"""
        return self.gemm.new_score_buffer(rows)

    def _fwd_bufs(self, B):
        return (self.H1[:B], self.MU[:B], self.LOGVAR[:B], self.EPSSTD[:B], self.ZS[:B],
                self.G1[:B], self.KLb[:B])

    def eval_factors(self, csr=None):
        """(user factors [rows][h + 1], item factors [n_items][h + 1]) whose inner products are the
        p-network logits at is_training = 0 of every CSR row (MultiVAE.py:186-206 with per-user inputs):
        user row = [g1(u) | 1], item row = [W_p1[i] | b_p1[i]].  The bias rides as the LAST factor
        column, so the k-ascending fmaf chain of the scoring kernels ends with fmaf(1, b, dot) = dot + b
        rounded once — bit for bit the `matmul + bias` of logits() — and the evaluation can take the
        factor path (pruned: no [users][I] logits slab, no bias pass, no mask pass)."""
        P = self.P
        csr = self.csr if csr is None else csr
        n, h = csr.n_rows, P["Wp1t"].shape[1]
        dev = P["Wp1t"].device
        z = P["Wp0"].shape[0]
        bufs = (torch.empty((n, h), device=dev), torch.empty((n, z), device=dev), torch.empty((n, z), device=dev),
                torch.empty((n, z), device=dev), torch.empty((n, z), device=dev), torch.empty((n, h), device=dev),
                torch.empty(n, device=dev))
        rows = torch.arange(n, dtype=torch.int32, device=dev)
        E.vae_encode(csr, rows, P["Wq0"], P["bq0"], P["Wq1"], P["bq1"], P["Wp0"], P["bp0"], self.act, 1.0, 0.0,
                     self.seed, self.t, bufs)
        users = torch.cat([bufs[5], torch.ones((n, 1), device=dev)], dim=1).contiguous()
        items = torch.cat([P["Wp1t"], P["bp1"].reshape(-1, 1)], dim=1).contiguous()
        return users, items

    def logits(self, rows, csr=None, out=None):
        """p_graph output for the given CSR rows at is_training=0, keep_prob=1 (MultiVAE.py:186-206
        feeds only input_ph).  Any number of rows (processed max_batch at a time); returns a
        [B][ld] score slab whose columns >= n_items are padding."""
        P, B = self.P, rows.numel()
        csr = self.csr if csr is None else csr
        if out is None:
            out = self.S if B <= self.max_batch else self.gemm.new_score_buffer(B)
        self.gemm.prepare(P["Wp1t"])
        for lo in range(0, B, self.max_batch):
            n = min(self.max_batch, B - lo)
            E.vae_encode(csr, rows[lo:lo + n], P["Wq0"], P["bq0"], P["Wq1"], P["bq1"], P["Wp0"],
                         P["bp0"], self.act, 1.0, 0.0, self.seed, self.t, self._fwd_bufs(n))
            S = self.gemm(self.G1[:n], None, out=out[lo:lo + n])
            E.add_row_bias(S, self.n_items, P["bp1"])
        return out[:B]

    def step(self, rows, anneal, keep=0.8, drop_given=None, eps_given=None, want_loss=True,
             apply=True):
        """One optimiser step on the users `rows` (int32 device tensor).  Leaves
        stats = [neg_ll, KL]; loss = neg_ll + anneal·KL + 2·reg_var (see loss()).
        apply=False stops after the gradients (self.G; the caller zeroes G["Wq0"] afterwards)."""
        P, G, B = self.P, self.G, rows.numel()
        if B > self.max_batch or B < 1:
            raise ValueError("batch size %d outside [1, %d]" % (B, self.max_batch))
        if self.native_step:
            # another learner than adam: the native call stops after the gradients, the update follows tensor by tensor
            E.vae_step_native(self, rows, anneal, keep, drop_given, eps_given, want_loss, apply and self.learner is None)
            self.last_anneal = float(anneal)
            if apply:
                if self.learner is not None:
                    self.learner.apply([(P[k], self.M[k], self.V[k], G[k], k == "Wq0") for k in self.NAMES])
                self.adam.advance()
                self.t += 1
            return
        E.vae_encode(self.csr, rows, P["Wq0"], P["bq0"], P["Wq1"], P["bq1"], P["Wp0"], P["bp0"],
                     self.act, keep, 1.0, self.seed, self.t, self._fwd_bufs(B),
                     drop_given=drop_given, eps_given=eps_given, h0val=self.h0val)
        if self.decoder == "fused":
            E.vae_decoder_fused(self.n_items, P["bp1"], self.csr, rows, self.G1[:B], P["Wp1t"], self.nll[:B],
                                G["Wp1t"], G["bp1"], self.dG1[:B], self.ws)
        else:
            self.gemm.prepare(P["Wp1t"])
            S = self.gemm(self.G1[:B], None, out=self.S)
            E.vae_decoder_loss_grad(S, self.n_items, P["bp1"], self.csr, rows, self.G1[:B], P["Wp1t"],
                                    self.nll[:B], G["Wp1t"], G["bp1"], self.dG1[:B], self.ws)
        E.vae_mid_backward(B, self.act, anneal, self.dG1[:B], self.G1[:B], self.H1[:B], self.MU[:B],
                           self.LOGVAR[:B], self.EPSSTD[:B], self.ZS[:B], P["Wp0"], P["Wq1"],
                           self.DA3[:B], self.DH2[:B], self.DA1[:B], G["Wp0"], G["bp0"], G["Wq1"],
                           G["bq1"], G["bq0"])
        E.vae_dwq0(self.csr, rows, self.h0val, self.DA1[:B], G["Wq0"])   # G["Wq0"] is zero here
        if want_loss:
            E.mean2_f32(self.nll[:B], self.KLb[:B], self.stats)
        if self.reg != 0.0:
            if want_loss:
                self.regsum.zero_()
            for k in ("Wq0", "Wq1", "Wp0", "Wp1t"):
                if want_loss:
                    E.sumsq_accumulate(P[k], self.regsum)
                E.axpy(2.0 * self.reg, P[k], G[k])
        self.last_anneal = float(anneal)
        if apply:
            self.apply_gradients()

    def gradient_tensors(self):
        """the gradients of the last step(apply=False), in NAMES order (replicas.MultiVAEReplicas sums them over ranks)"""
        return [self.G[k] for k in self.NAMES]

    def set_gradient_tensors(self, tensors):
        for k, t in zip(self.NAMES, tensors):
            self.G[k] = t

    def apply_gradients(self):
        """the update half of a step: the learner on self.G (dW_q0, accumulated by row, is cleared behind it)"""
        tensors = [(self.P[k], self.M[k], self.V[k], self.G[k], k == "Wq0") for k in self.NAMES]
        if self.learner is None:
            E.adam_dense_multi(tensors, self.adam)
        else:
            self.learner.apply(tensors)
        self.adam.advance()
        self.t += 1

    def loss(self):
        """Host read of the last step's neg-ELBO (syncs)."""
        neg_ll, kl = (float(x) for x in self.stats.cpu())
        reg_var = self.reg * float(self.regsum.item()) / 2.0 if self.reg != 0.0 else 0.0
        return neg_ll + self.last_anneal * kl + 2.0 * reg_var, neg_ll, kl
