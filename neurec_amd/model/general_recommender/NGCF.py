"""NGCF on the HIP engine (alg_type=ngcf, the reference's default).

Paper: Xiang Wang, Xiangnan He, Meng Wang, Fuli Feng, Tat-Seng Chua, "Neural Graph Collaborative
Filtering", SIGIR 2019.  Plugin-compatible with model/general_recommender/NGCF.py: same
constructor, config keys (conf/NGCF.properties), training loop and log lines.  Deviations from
the reference are limited to what SURVEY.md H6 lists as its quirks and are stated, not hidden:
  * the adjacency is built sparsely (the reference densifies the U×I train matrix first);
  * message dropout stays active at evaluation, as in the reference (no training flag there);
  * alg_type ngcf / gcn / gcmc, node dropout and every learner of util/learner.py are built (r05); the shipped
    defaults (ngcf, adam, 16 / [16, 16], no node dropout) run on the fused register-resident engine, everything else on
    the width-generic one.
"""
from time import time

import numpy as np

from ...data import PairwiseSampler
from ...graph import ngcf_adjacency, transpose_csr
from ...util import timer
from ...util.tool import get_initializer
from ..AbstractRecommender import AbstractRecommender
from ._common import predict_scores


class NGCF(AbstractRecommender):
    def __init__(self, sess, dataset, conf):
        super(NGCF, self).__init__(dataset, conf)
        self.learning_rate = conf["learning_rate"]
        self.learner = conf["learner"]
        self.batch_size = conf["batch_size"]
        self.emb_dim = conf["embedding_size"]
        self.weight_size = conf["layer_size"]
        self.n_layers = len(self.weight_size)
        self.num_epochs = conf["epochs"]
        self.reg = conf["reg"]
        self.adj_type = conf["adj_type"]
        self.alg_type = conf["alg_type"]
        self.node_dropout_flag = conf["node_dropout_flag"]
        self.node_dropout_ratio = conf["node_dropout_ratio"]
        self.mess_dropout_ratio = conf["mess_dropout_ratio"]
        self.embed_init_method = conf["embed_init_method"]
        self.weight_init_method = conf["weight_init_method"]
        self.stddev = conf["stddev"]
        self.verbose = conf["verbose"]
        self.dataset = dataset
        self.num_users = dataset.num_users
        self.num_items = dataset.num_items
        self.norm_adj = ngcf_adjacency(dataset.train_matrix, self.adj_type)
        self.logger.info({"plain": "use the plain adjacency matrix",
                          "norm": "use the normalized adjacency matrix",
                          "gcmc": "use the gcmc adjacency matrix"}.get(self.adj_type,
                                                                       "use the mean adjacency matrix"))
        self.n_nonzero_elems = self.norm_adj.count_nonzero()
        self.sess = sess
        self.engine = None
        self._final = None

    def build_graph(self):
        from ... import engine as E
        from ...trainer import NGCFEngine
        if self.alg_type not in ("ngcf", "gcn", "gcmc"):
            raise ValueError("alg_type must be ngcf, gcn or gcmc")            # NGCF.py:67-74 silently builds nothing
        if str(self.learner).lower() not in ("adam",) + E.DenseLearner.KINDS:
            raise ValueError("please select a suitable optimizer")             # util/learner.py:15-16
        sizes = [self.emb_dim] + list(self.weight_size)
        blocks = sizes[1:] if self.alg_type == "gcmc" else sizes               # gcmc concatenates the dense layers only
        if any(not 1 <= s <= 256 for s in sizes) or sum(blocks) > 256:
            raise NotImplementedError("NGCF layer widths 1..256 with a concatenated width <= 256 are built "
                                      "(embedding_size + sum(layer_size) = %d)" % sum(sizes))
        e_init = get_initializer(self.embed_init_method, self.stddev, seed=2017)
        w_init = get_initializer(self.weight_init_method, self.stddev, seed=2018)
        table = np.concatenate([e_init([self.num_users, self.emb_dim]),
                                e_init([self.num_items, self.emb_dim])])
        self.logger.info("using xavier initialization")
        weights = []
        for k in range(self.n_layers):
            layer = [w_init([sizes[k], sizes[k + 1]]), w_init([1, sizes[k + 1]]),
                     w_init([sizes[k], sizes[k + 1]]), w_init([1, sizes[k + 1]])]     # W_gc, b_gc, W_bi, b_bi (NGCF.py:271-279)
            if self.alg_type == "gcmc":                                               # W_mlp, b_mlp (NGCF.py:281-284)
                layer[2:] = [w_init([sizes[k], sizes[k + 1]]), w_init([1, sizes[k + 1]])]
            weights.append(tuple(layer))
        node_dropout = float(self.node_dropout_ratio) if (self.node_dropout_flag is True and self.alg_type == "ngcf") else 0.0
        args = (self.norm_adj, transpose_csr(self.norm_adj), self.num_users, self.num_items, table, weights,
                self.learning_rate, self.reg, self.mess_dropout_ratio, self.batch_size)
        from ... import parallel
        self.comm = parallel.get_comm()
        self.dp_mode = None
        if self.comm.active and self.alg_type == "ngcf" and node_dropout == 0.0 and str(self.learner).lower() == "adam":
            # one rank of several (python -m torch.distributed.run -m neurec_amd.main): node rows sharded over the ranks,
            # the tiny layer weights replicated, `batch_size` = the GLOBAL batch (sharded_ngcf.ShardedNGCF).  The other
            # alg_types / learners / node dropout train the whole model on every rank (same seeds, same tables)
            from ...sharded_ngcf import ShardedNGCF
            self.dp_mode = "rowshard"
            self.engine = ShardedNGCF(self.comm, *args)
            return
        if all(s == 16 for s in sizes) and self.alg_type == "ngcf" and node_dropout == 0.0:
            # the shipped configuration: fused, register-resident layer kernels
            self.engine = NGCFEngine(*args, learner=self.learner)
            return
        # any other widths / alg_type / node dropout: SpMM + fp32-MFMA GEMM + row-wise kernels
        from ...ngcf_wide import NGCFWideEngine
        self.logger.info("embedding_size=%d layer_size=%s alg_type=%s%s runs on the width-generic NGCF engine"
                         % (self.emb_dim, list(self.weight_size), self.alg_type,
                            " node_dropout=%g" % node_dropout if node_dropout else ""))
        self.engine = NGCFWideEngine(*args, learner=self.learner, alg_type=self.alg_type, node_dropout=node_dropout)

    def train_model(self):
        import torch
        self.logger.info(self.evaluator.metrics_info())
        data_iter = PairwiseSampler(self.dataset, neg_num=1, batch_size=self.batch_size,
                                    shuffle=True, as_tensors=True)
        losses = torch.zeros((max(len(data_iter), 1), 2), device="cuda")
        comm = self.comm
        for epoch in range(1, self.num_epochs + 1):
            training_start_time = time()
            num_training_instances = len(data_iter)
            n = 0
            for batch in data_iter:
                bat_users, bat_items_pos, bat_items_neg = batch
                if self.dp_mode == "rowshard":           # this rank's slice of the global batch (every rank draws the same stream)
                    nb = bat_users.numel()
                    lo, hi = (nb * comm.rank) // comm.world, (nb * (comm.rank + 1)) // comm.world
                    self.engine.step(bat_users[lo:hi], bat_items_pos[lo:hi], bat_items_neg[lo:hi], losses[n])
                else:
                    self.engine.step(bat_users, bat_items_pos, bat_items_neg, losses[n], plan=batch.plan)
                n += 1
            if self.dp_mode == "rowshard":
                comm.allreduce_sum_(losses)               # the ranks' per-step sums added: the global batch's loss
            total_loss = 0.0
            for a, b in losses[:n].cpu().numpy():
                total_loss += np.float32(a) + np.float32(b)
            self.logger.info("[iter %d : loss : %f, time: %f]" % (epoch, total_loss / num_training_instances,
                                                                 time() - training_start_time))
            if epoch % self.verbose == 0:
                self.logger.info("epoch %d:\t%s" % (epoch, self.evaluate()))

    @timer
    def evaluate(self):
        eu, ei = self.engine.final_embeddings()
        self._final = (eu.contiguous(), ei.contiguous())
        return self.evaluator.evaluate(self)

    def get_eval_factors(self):
        if self._final is None:
            eu, ei = self.engine.final_embeddings()
            self._final = (eu.contiguous(), ei.contiguous())
        return self._final

    def predict(self, user_ids, candidate_items_userids=None):
        eu, ei = self.get_eval_factors()
        return predict_scores(eu, ei, user_ids, candidate_items_userids)
