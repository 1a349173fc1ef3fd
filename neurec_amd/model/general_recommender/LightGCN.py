"""LightGCN on the HIP engine.

Paper: LightGCN: Simplifying and Powering Graph Convolution Network for Recommendation
(He, Deng, Wang, Li, Zhang, Wang).  Plugin-compatible with
model/general_recommender/LightGCN.py: same constructor, config keys
(conf/LightGCN.properties), adjacency options, training loop and log lines.  Every training
step re-propagates the whole graph, as the reference's loss graph does; propagation is the CSR
SpMM kernel, the BPR head / its gradient / dense Adam are HIP kernels, triplets are sampled on
the device, and evaluation ranks from the propagated tables without leaving HBM.
"""
import numpy as np

from ...data import PairwiseSampler
from ...graph import is_symmetric, lightgcn_adjacency, transpose_csr
from ...util import timer
from ..AbstractRecommender import AbstractRecommender
from ._common import predict_scores


class LightGCN(AbstractRecommender):
    def __init__(self, sess, dataset, config):
        super(LightGCN, self).__init__(dataset, config)
        self.lr = config["lr"]
        self.reg = config["reg"]
        self.emb_dim = config["embed_size"]
        if not (isinstance(self.emb_dim, int) and 1 <= self.emb_dim <= 256):
            # fail here, by name, not as an opaque context error inside the native step.  Widths other than
            # 16 / 32 / 64 / 128 / 256 run zero-padded to the next built one (trainer.LightGCNEngine); beyond 128 columns
            # the evaluation scores through the general fp32-MFMA GEMM (engine.ScoreGemmWide: the same k-ascending
            # chain per score, materialised path) instead of the tile-maxima scoring loop
            raise NotImplementedError("the HIP LightGCN plugin is built for embed_size 1..256; got %r" % (self.emb_dim,))
        self.batch_size = config["batch_size"]
        self.epochs = config["epochs"]
        self.n_layers = config["n_layers"]
        self.dataset = dataset
        self.n_users, self.n_items = self.dataset.num_users, self.dataset.num_items
        self.user_pos_train = self.dataset.get_user_train_dict(by_time=False)
        self.all_users = list(self.user_pos_train.keys())
        self.adj_type = config["adj_type"]
        self.dp_mode_conf = config["dp_mode"] if "dp_mode" in config else None     # multi-GPU runs only (build_graph)
        self.norm_adj = self.create_adj_mat(config["adj_type"])
        self.sess = sess                      # unused: there is no TensorFlow session
        self.engine = None
        self._final = None

    @timer
    def create_adj_mat(self, adj_type):
        user_list, item_list = self.dataset.get_train_interactions()
        # tf_order: the rows in the order the reference hands them to TensorFlow (`gcmc`: descending columns —
        # its single scipy product leaves them so, LightGCN.py:48); the kernels sum a row in storage order
        adj = lightgcn_adjacency(user_list, item_list, self.n_users, self.n_items, adj_type, tf_order=True)
        print({"plain": "use the plain adjacency matrix", "norm": "use the normalized adjacency matrix",
               "gcmc": "use the gcmc adjacency matrix", "pre": "use the pre adjcency matrix"}
              .get(adj_type, "use the mean adjacency matrix"))
        return adj

    def build_graph(self):
        from ...trainer import LightGCNEngine
        rng = np.random.RandomState(2017)                 # main.py:12
        n = self.n_users + self.n_items
        # tf.contrib.layers.xavier_initializer() on [n_users, d] and [n_items, d] (LightGCN.py:87-89)
        lim_u = np.sqrt(6.0 / (self.n_users + self.emb_dim))
        lim_i = np.sqrt(6.0 / (self.n_items + self.emb_dim))
        table = np.empty((n, self.emb_dim), dtype=np.float32)
        table[:self.n_users] = rng.uniform(-lim_u, lim_u, (self.n_users, self.emb_dim))
        table[self.n_users:] = rng.uniform(-lim_i, lim_i, (self.n_items, self.emb_dim))
        adj_t = None if is_symmetric(self.norm_adj) else transpose_csr(self.norm_adj)
        from ... import parallel
        self.comm = comm = parallel.get_comm()
        self.dp_mode = None
        if comm.active:
            # one rank of several: `batch_size` stays the GLOBAL batch of a step (the run is the one-GPU run of the same
            # configuration, partitioned).  --dp_mode=colshard (default where embed_size divides by the ranks): every
            # rank holds embed_size / N columns of every row and steps on the whole batch, one all-gather of 12 B per
            # triplet per step (colshard.py); --dp_mode=rowshard: tables row-sharded, a rank steps on its slice of the
            # batch, rows travel by all-to-all (sharded.py: north_star's partition, the form for tables one GPU
            # cannot hold) — bit-identical to the one-GPU run
            mode = self.dp_mode_conf or ("colshard" if self.emb_dim % comm.world == 0 else "rowshard")
            if mode not in ("colshard", "rowshard"):
                raise ValueError("dp_mode must be 'colshard' or 'rowshard', got %r" % (mode,))
            self.dp_mode = mode
            if mode == "colshard":
                from ...colshard import ColumnShardedLightGCN
                self.engine = ColumnShardedLightGCN(comm, self.norm_adj, self.n_users, self.n_items, table, self.n_layers,
                                                    self.lr, self.reg, self.batch_size, adj_t_csr=adj_t, keep_order=True)
            else:
                from ...sharded import ShardedLightGCN
                if self.adj_type == "gcmc":
                    raise NotImplementedError("dp_mode=rowshard sums a row in ascending column order; adj_type=gcmc's "
                                              "descending order is the one-GPU / colshard engines'")
                self.engine = ShardedLightGCN(comm, self.norm_adj, self.n_users, self.n_items, table, self.n_layers,
                                              self.lr, self.reg, self.batch_size)
            return
        self.engine = LightGCNEngine(self.norm_adj, self.n_users, self.n_items, table,
                                     self.n_layers, self.lr, self.reg, self.batch_size,
                                     adj_t_csr=adj_t, keep_order=True)

    def train_model(self):
        import torch
        data_iter = PairwiseSampler(self.dataset, neg_num=1, batch_size=self.batch_size,
                                    shuffle=True, as_tensors=True)
        loss2 = torch.zeros(2, device="cuda")
        self.logger.info(self.evaluator.metrics_info())
        comm = self.comm
        for epoch in range(self.epochs):
            for batch in data_iter:
                bat_users, bat_pos_items, bat_neg_items = batch
                if self.dp_mode == "rowshard":
                    # every rank draws the same epoch stream (a counter-based generator: same seed, same triplets) and
                    # steps on ITS contiguous slice of each global batch; the owners add the gradient rows in the
                    # global batch's order: the one-GPU step on the whole batch, bit for bit
                    n = bat_users.numel()
                    lo, hi = (n * comm.rank) // comm.world, (n * (comm.rank + 1)) // comm.world
                    self.engine.step(bat_users[lo:hi], bat_pos_items[lo:hi], bat_neg_items[lo:hi], loss2)
                else:
                    self.engine.step(bat_users, bat_pos_items, bat_neg_items, loss2, plan=batch.plan)
            result = self.evaluate_model()
            self.logger.info("epoch %d:\t%s" % (epoch, result))

    def _snapshot(self):
        # the reference's assign_opt: snapshot the propagated tables once per evaluation.  Row-sharded: a rank keeps
        # ITS user rows and gets the whole item table (one all-gather of the item blocks)
        if self.dp_mode == "rowshard":
            eu, ei = self.engine.eval_factors()
        else:
            eu, ei = self.engine.final_embeddings()
        self._final = (eu.contiguous(), ei.contiguous())

    def evaluate_model(self):
        self._snapshot()
        return self.evaluator.evaluate(self)

    def get_eval_factors(self):
        if self._final is None:
            self._snapshot()
        return self._final

    def eval_user_range(self):
        """row-sharded run: the users whose rows get_eval_factors()[0] holds (row r = user lo + r); else None"""
        return (self.engine.ulo, self.engine.uhi) if self.dp_mode == "rowshard" else None

    def predict(self, users, candidate_items=None):
        if self.dp_mode == "rowshard":
            eu, ei = self.engine.final_embeddings()       # the plugin contract scores ANY user: the whole table, gathered
            return predict_scores(eu.contiguous(), ei.contiguous(), users, candidate_items)
        eu, ei = self.get_eval_factors()
        return predict_scores(eu, ei, users, candidate_items)
