"""BPR-MF on the HIP engine.

Reference: Steffen Rendle et al., "BPR: Bayesian Personalized Ranking from Implicit Feedback."
UAI 2009.  Plugin-compatible with model/general_recommender/MF.py: same constructor, config
keys (conf/MF.properties), training loop, log lines and `predict` contract; the per-batch
`sess.run((loss, optimizer))` is one fused gather/BPR/scatter kernel plus two TF-semantics Adam
sweeps on tables that never leave HBM, and the triplets are sampled on the device.
"""
from time import time

import numpy as np

from ...data import PairwiseSampler, PointwiseSampler
from ...util import timer
from ...util.tool import get_initializer
from ..AbstractRecommender import AbstractRecommender
from ._common import predict_scores


class MF(AbstractRecommender):
    def __init__(self, sess, dataset, conf):
        super(MF, self).__init__(dataset, conf)
        self.learning_rate = conf["learning_rate"]
        self.embedding_size = conf["embedding_size"]
        self.learner = conf["learner"]
        self.loss_function = conf["loss_function"]
        self.is_pairwise = conf["is_pairwise"]
        self.num_epochs = conf["epochs"]
        self.reg_mf = conf["reg_mf"]
        self.batch_size = conf["batch_size"]
        self.verbose = conf["verbose"]
        self.num_negatives = conf["num_negatives"]
        self.init_method = conf["init_method"]
        self.stddev = conf["stddev"]
        self.dataset = dataset
        self.num_users = dataset.num_users
        self.num_items = dataset.num_items
        self.sess = sess                      # unused: there is no TensorFlow session
        self.dp_mode_conf = conf["dp_mode"] if "dp_mode" in conf else None      # multi-GPU runs only (build_graph)
        self.engine = None

    def build_graph(self):
        from ...trainer import GeneralMFEngine, MFEngine
        init = get_initializer(self.init_method, self.stddev, seed=2017)   # main.py:12
        users = init([self.num_users, self.embedding_size])
        items = init([self.num_items, self.embedding_size])
        self._fast = (self.is_pairwise is True and str(self.loss_function).lower() == "bpr"
                      and str(self.learner).lower() == "adam")
        from ... import parallel
        self.comm = parallel.get_comm()
        self.dp_mode = None
        if self.comm.active and self._fast and (self.dp_mode_conf or "rowshard") == "rowshard":
            # one rank of several (python -m torch.distributed.run -m neurec_amd.main): both tables row-sharded with
            # their Adam moments, `batch_size` stays the GLOBAL batch — a rank steps on its slice of it, ids / rows /
            # gradient rows travel by all-to-all, the owners add the gradient rows in the global batch's order
            # (sharded.ShardedMF: the one-GPU step on the whole batch, bit for bit).  Other losses / learners, or
            # --dp_mode=replicated: every rank trains the whole model (same seeds, same tables), evaluation is shared
            from ...sharded import ShardedMF
            self.dp_mode = "rowshard"
            self.engine = ShardedMF(self.comm, users, items, self.learning_rate, self.reg_mf, self.batch_size)
            return
        if self._fast:                         # conf/MF.properties as shipped: one native call per step
            self.engine = MFEngine(users, items, self.learning_rate, self.reg_mf, self.batch_size)
        else:                                  # the other losses / optimisers of util/learner.py
            self.engine = GeneralMFEngine(users, items, self.learning_rate, self.reg_mf,
                                          self.batch_size, loss=self.loss_function,
                                          pairwise=self.is_pairwise is True, learner=self.learner)

    # ---------- training process -------
    def train_model(self):
        import torch
        self.logger.info(self.evaluator.metrics_info())
        if self.dp_mode == "rowshard":
            return self._train_rowshard()
        dev = self.engine.P.device
        if self.is_pairwise is True:           # MF.py:88-93
            data_iter = PairwiseSampler(self.dataset, neg_num=1, batch_size=self.batch_size,
                                        shuffle=True, as_tensors=True)
        else:
            data_iter = PointwiseSampler(self.dataset, neg_num=self.num_negatives,
                                         batch_size=self.batch_size, shuffle=True)
        losses = torch.zeros((max(len(data_iter), 1), 2), device=dev)
        for epoch in range(1, self.num_epochs + 1):
            training_start_time = time()
            n = 0
            if self._fast and self.engine.max_batch >= self.batch_size:
                # conf/MF.properties as shipped: the whole batch loop of the epoch is one native call
                users, items, negs, plans = data_iter.epoch_stream()
                n = self.engine.run_batches(users, items, negs, self.batch_size, losses, plans)
            for batch in (() if n else data_iter):
                bat_users, bat_items, bat_third = batch
                if self.is_pairwise is not True:      # host lists from the pointwise iterator
                    bat_users = torch.tensor(bat_users, dtype=torch.int32, device=dev)
                    bat_items = torch.tensor(bat_items, dtype=torch.int32, device=dev)
                    bat_third = torch.tensor(bat_third, dtype=torch.float32, device=dev)
                if self._fast:                        # the batch's plan came with it from the sampler
                    self.engine.step(bat_users, bat_items, bat_third, losses[n], plan=batch.plan,
                                     next_plan=batch.next_plan)
                else:
                    self.engine.step(bat_users, bat_items, bat_third, losses[n])
                n += 1
            per_step = losses[:n].cpu().numpy()           # one D2H copy per epoch
            total_loss = 0.0
            for a, b in per_step:                          # `total_loss += loss`, MF.py:102
                total_loss += np.float32(a) + np.float32(b)
            self.logger.info("[iter %d : loss : %f, time: %f]" % (epoch, total_loss / len(data_iter),
                                                                 time() - training_start_time))
            if epoch % self.verbose == 0:
                self.logger.info("epoch %d:\t%s" % (epoch, self.evaluate()))

    def _train_rowshard(self):
        """train_model as one rank of a row-sharded run: the same epoch stream on every rank, a rank's slice of each
        global batch, the logged loss = the ranks' per-step sums added (one all-reduce per epoch)"""
        import torch
        comm = self.comm
        data_iter = PairwiseSampler(self.dataset, neg_num=1, batch_size=self.batch_size, shuffle=True, as_tensors=True)
        losses = torch.zeros((max(len(data_iter), 1), 2), device="cuda")
        for epoch in range(1, self.num_epochs + 1):
            training_start_time = time()
            n = 0
            for batch in data_iter:
                bat_users, bat_items, bat_negs = batch
                nb = bat_users.numel()
                lo, hi = (nb * comm.rank) // comm.world, (nb * (comm.rank + 1)) // comm.world
                self.engine.step(bat_users[lo:hi], bat_items[lo:hi], bat_negs[lo:hi], losses[n])
                n += 1
            comm.allreduce_sum_(losses)
            per_step = losses[:n].cpu().numpy()
            total_loss = 0.0
            for a, b in per_step:                          # `total_loss += loss`, MF.py:102
                total_loss += np.float32(a) + np.float32(b)
            self.logger.info("[iter %d : loss : %f, time: %f]" % (epoch, total_loss / len(data_iter),
                                                                 time() - training_start_time))
            if epoch % self.verbose == 0:
                self.logger.info("epoch %d:\t%s" % (epoch, self.evaluate()))

    @timer
    def evaluate(self):
        self._factors = self.engine.eval_factors() if self.dp_mode == "rowshard" else None
        return self.evaluator.evaluate(self)

    def get_eval_factors(self):
        """Device tables for the evaluator's on-GPU factor path (row-sharded: this rank's user rows + every item row)."""
        if self.dp_mode == "rowshard":
            if getattr(self, "_factors", None) is None:
                self._factors = self.engine.eval_factors()
            return self._factors
        return self.engine.P, self.engine.Q

    def eval_user_range(self):
        """row-sharded run: the users whose rows get_eval_factors()[0] holds (row r = user lo + r); else None"""
        return self.engine.part.users_of(self.engine.rank) if self.dp_mode == "rowshard" else None

    def predict(self, user_ids, candidate_items=None):
        if self.dp_mode == "rowshard":
            P, Q = self.engine.tables()                   # the plugin contract scores ANY user: the tables, gathered
            return predict_scores(P.contiguous(), Q.contiguous(), user_ids, candidate_items)
        return predict_scores(self.engine.P, self.engine.Q, user_ids, candidate_items)
