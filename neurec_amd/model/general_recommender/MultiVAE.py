"""Mult-VAE on the HIP engine.

Paper: Dawen Liang, Rahul G. Krishnan, Matthew D. Hoffman, Tony Jebara, "Variational
Autoencoders for Collaborative Filtering", WWW 2018.  Plugin-compatible with
model/general_recommender/MultiVAE.py: same constructor, config keys (conf/MultiVAE.properties),
epoch loop (user permutation, int(num_users/batch_size) full batches, KL annealing over
update_count, keep_prob 0.8) and log lines.

What is mirrored on purpose (stated rather than hidden):
  * only int(num_users / batch_size) batches run per epoch — the tail users are skipped
    (MultiVAE.py:147);
What is NOT mirrored by default:
  * the reference's `predict` feeds ONE rating row that is never cleared between the users of a
    call, so user k of a predict batch is scored on the union of the histories of users 0..k
    (MultiVAE.py:186-206): the metrics then depend on how test users are batched and other users'
    interactions leak into every score.  Here each user is scored on their own history (the model
    that was trained).  `--reference_predict_rows=True` (or `predict_accumulates_rows = True`) is
    the explicit compatibility switch that reproduces the reference's rows; it is logged when on.
Shapes: p_dim may have any number of layers and any widths (conf/MultiVAE.properties:3 lists [200, 600] and
[200] next to the shipped [16, 32]).  The shipped two-layer shape with z <= 16, h <= 32 runs on the register-resident
kernels (trainer.MultiVAEEngine); every other shape on the width-generic kernels + the fp32-MFMA GEMM
(vae_wide.MultiVAEWideEngine), whose evaluation goes through predict()'s logits slab.
"""
from time import time

import numpy as np

from ...util import timer
from ...util.tool import csr_to_user_dict, get_initializer
from ..AbstractRecommender import AbstractRecommender


class MultiVAE(AbstractRecommender):
    predict_accumulates_rows = False

    def __init__(self, sess, dataset, conf):
        super(MultiVAE, self).__init__(dataset, conf)
        self.learning_rate = conf["learning_rate"]
        self.learner = conf["learner"]
        self.batch_size = conf["batch_size"]
        self.act = conf["activation"]
        self.reg = conf["reg"]
        self.num_epochs = conf["epochs"]
        self.anneal_cap = conf["anneal_cap"]
        self.total_anneal_steps = conf["total_anneal_steps"]
        self.weight_init_method = conf["weight_init_method"]
        self.bias_init_method = conf["bias_init_method"]
        self.stddev = conf["stddev"]
        self.verbose = conf["verbose"]
        self.dataset = dataset
        self.num_users = dataset.num_users
        self.num_items = dataset.num_items
        self.p_dims = list(conf["p_dim"]) + [self.num_items]
        self.q_dims = self.p_dims[::-1]
        self.dims = self.q_dims + self.p_dims[1:]
        self.train_dict = csr_to_user_dict(dataset.train_matrix)
        self.sess = sess
        self.engine = None
        if "reference_predict_rows" in conf:
            self.predict_accumulates_rows = bool(conf["reference_predict_rows"])
        if self.predict_accumulates_rows:
            self.logger.info("reference_predict_rows=True: predict() accumulates the rating row over "
                             "the users of a call, as MultiVAE.py:186-206 does")
        else:
            self.logger.info("reference_predict_rows=False (default): every user is scored on their OWN "
                             "history; the reference's predict() never clears its rating row between the users "
                             "of a call (MultiVAE.py:186-206), so its evaluation metrics differ from these — "
                             "pass --reference_predict_rows=True to reproduce them")

    def build_graph(self):
        from ... import engine as E
        from ...trainer import MultiVAEEngine
        if str(self.learner).lower() not in ("adam",) + E.DenseLearner.KINDS:
            raise ValueError("please select a suitable optimizer")               # util/learner.py:15-16
        if self.act not in E.VAE_ACTS:
            raise NotImplementedError("activation %r is not built (tanh/sigmoid/relu/identity)" % self.act)
        w_init = get_initializer(self.weight_init_method, self.stddev, seed=2017)
        b_init = get_initializer(self.bias_init_method, self.stddev, seed=2018)
        train = self.dataset.train_matrix.tocsr().astype(np.float32)
        train.sort_indices()
        self._train_csr = train
        n = self.num_items
        narrow = len(self.p_dims) == 3 and 1 <= self.p_dims[0] <= 16 and 1 <= self.p_dims[1] <= 32
        if narrow:
            z, h, _ = self.p_dims
            params = {
                "Wq0": w_init([n, h]), "bq0": b_init([h]),
                "Wq1": w_init([h, 2 * z]), "bq1": b_init([2 * z]),
                "Wp0": w_init([z, h]), "bp0": b_init([h]),
                # TF variable weight_p_1to2 is [h, n]; the engine keeps it item-major
                "Wp1t": np.ascontiguousarray(w_init([h, n]).T), "bp1": b_init([n]),
            }
            self.engine = MultiVAEEngine(E.DeviceCSR.from_scipy(train), n, params, self.learning_rate,
                                         self.reg, self.act, max(self.batch_size, 1), learner=self.learner)
            return
        # any other p_dim: the variables in the order the reference creates them (MultiVAE.py:46-71: all of q, then p)
        from ...vae_wide import MultiVAEWideEngine
        Wq, bq, Wp, bp = [], [], [], []
        for i, (d_in, d_out) in enumerate(zip(self.q_dims[:-1], self.q_dims[1:])):
            if i == len(self.q_dims[:-1]) - 1:
                d_out *= 2                                     # mean and log-variance (MultiVAE.py:50-52)
            Wq.append(w_init([d_in, d_out]))
            bq.append(b_init([d_out]))
        for d_in, d_out in zip(self.p_dims[:-1], self.p_dims[1:]):
            Wp.append(w_init([d_in, d_out]))
            bp.append(b_init([d_out]))
        self.engine = MultiVAEWideEngine(E.DeviceCSR.from_scipy(train), n, Wq, bq, Wp, bp, self.learning_rate,
                                         self.reg, self.act, max(self.batch_size, 1), learner=self.learner)
        self.logger.info("p_dim=%s runs on the width-generic Mult-VAE engine (fp32 matrix-core GEMMs for the item layer)"
                         % (list(self.p_dims[:-1]),))

    # ------------------------------------------------------------------ training
    def train_model(self):
        import torch
        from ... import parallel
        update_count = 0.0
        self.logger.info(self.evaluator.metrics_info())
        dev = self.engine.stats.device
        comm = parallel.get_comm()
        replicas = None
        if comm.active and self.batch_size % comm.world == 0:
            # one rank of several: data-parallel replicas — every rank encodes / decodes ITS share of each
            # global batch of `batch_size` users, one all-reduce of the flat gradient buffer per step, the same dense
            # update everywhere (replicas.MultiVAEReplicas); the permutation below is the same on every rank (numpy
            # seed 2018, main.py:10).  A batch size the ranks do not divide trains the whole batch on every rank
            from ...replicas import MultiVAEReplicas
            replicas = MultiVAEReplicas(comm, self.engine, decorrelate=True)
        n_batches = int(self.num_users / self.batch_size)
        for epoch in range(1, self.num_epochs + 1):
            random_perm_doc_idx = np.random.permutation(self.num_users).astype(np.int32)
            perm_dev = torch.from_numpy(random_perm_doc_idx).to(dev)
            training_start_time = time()
            num_training_instances = self.num_users
            stats = torch.zeros((max(n_batches, 1), 2), dtype=torch.float32, device=dev)
            anneals, reg_total = [], 0.0
            for num_batch in range(n_batches):
                rows = perm_dev[num_batch * self.batch_size:(num_batch + 1) * self.batch_size]
                if self.total_anneal_steps > 0:
                    anneal = min(self.anneal_cap, 1. * update_count / self.total_anneal_steps)
                else:
                    anneal = self.anneal_cap
                if replicas is not None:
                    replicas.step(rows.contiguous(), anneal, keep=0.8)
                else:
                    self.engine.step(rows.contiguous(), anneal, keep=0.8)
                stats[num_batch].copy_(self.engine.stats)
                if self.reg != 0.0:                    # 2·reg_var term (host read; reg defaults to 0)
                    loss, neg_ll, kl = self.engine.loss()
                    reg_total += loss - neg_ll - anneal * kl
                anneals.append(anneal)
                update_count += 1
            if replicas is not None:                   # the ranks' batch means averaged: the global batch's means
                comm.allreduce_sum_(stats)
                stats /= comm.world
            total_loss = reg_total
            for (neg_ll, kl), anneal in zip(stats[:len(anneals)].cpu().numpy(), anneals):
                total_loss += float(neg_ll) + anneal * float(kl)
            self.logger.info("[iter %d : loss : %f, time: %f]" % (epoch, total_loss / num_training_instances,
                                                                 time() - training_start_time))
            if epoch % self.verbose == 0:
                self.logger.info("epoch %d:\t%s" % (epoch, self.evaluate()))

    @timer
    def evaluate(self):
        return self.evaluator.evaluate(self)

    def get_eval_factors(self):
        """Device factor tables for the evaluator's on-GPU path ([g1(u) | 1], [W_p1 | b_p1]: their inner
        products are predict()'s logits bit for bit) — only with per-user inputs; the reference's
        accumulating predict rows depend on how the test users are batched, so that mode returns None and
        is scored through predict()."""
        if self.predict_accumulates_rows or not hasattr(self.engine, "eval_factors"):
            return None
        return self.engine.eval_factors()

    # ------------------------------------------------------------------ inference
    def _predict_rows(self, user_ids):
        """CSR of the input rows of one predict call (see the module docstring)."""
        from ... import engine as E
        train = self._train_csr
        if not self.predict_accumulates_rows:
            return None, np.asarray(user_ids, dtype=np.int32)
        seen = np.zeros(self.num_items, dtype=bool)
        indptr = np.zeros(len(user_ids) + 1, dtype=np.int64)
        chunks = []
        for k, u in enumerate(user_ids):
            seen[train.indices[train.indptr[u]:train.indptr[u + 1]]] = True
            idx = np.flatnonzero(seen).astype(np.int32)
            chunks.append(idx)
            indptr[k + 1] = indptr[k] + len(idx)
        indices = np.concatenate(chunks) if chunks else np.zeros(0, np.int32)
        return E.DeviceCSR(indptr, indices, self.num_items), np.arange(len(user_ids), dtype=np.int32)

    def predict(self, user_ids, candidate_items_user_ids=None):
        """Logits of the p-network (self.h in the reference) for each user; full-rank mode returns a
        [B, num_items] device tensor view, candidate mode a list of per-user numpy arrays."""
        import torch
        user_ids = list(user_ids)
        csr, rows = self._predict_rows(user_ids)
        S = self.engine.logits(torch.from_numpy(rows).to(self.engine.stats.device), csr=csr,
                               out=self.engine.gemm_out(len(user_ids)))
        ratings = S[:, :self.num_items]
        if candidate_items_user_ids is not None:
            host = ratings.cpu().numpy()
            return [host[k, items] for k, items in enumerate(candidate_items_user_ids)]
        return ratings
