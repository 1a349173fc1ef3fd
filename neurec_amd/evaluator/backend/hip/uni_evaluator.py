"""UniEvaluator on the HIP engine — the driver of evaluator/backend/cpp/uni_evaluator.py:101-157.

Two entrances:
  * factor fast path — a model that exposes `get_eval_factors()` -> (user_table, item_table)
    device tensors is scored, masked, ranked and measured entirely on the GPU
    (trainer.FullRankEvaluator); nothing but M*K numbers per user leaves HBM.
  * score-matrix path — any plugin whose `predict(users, candidate_items)` returns a [B, N]
    array or a list of per-user arrays (the reference contract) is evaluated batch by batch:
    scores go to the device once, the -inf train mask, top-K and metrics run there.
Both produce the reference's per-user float32 metric rows, the float32 mean over users and
the same "%.8f" string.
"""
import numpy as np

from ....util.data_iterator import DataIterator
from ....util.tool import pad_sequences, typeassert
from .hip_evaluator import HIPEvaluator, float_type

metric_dict = {"Precision": 1, "Recall": 2, "MAP": 3, "NDCG": 4, "MRR": 5}
re_metric_dict = {value: key for key, value in metric_dict.items()}


class UniEvaluator(HIPEvaluator):
    @typeassert(user_train_dict=dict, user_test_dict=(dict, None.__class__))
    def __init__(self, user_train_dict, user_test_dict, user_neg_test=None, metric=None,
                 top_k=50, batch_size=1024, num_thread=8):
        super(UniEvaluator, self).__init__()
        if metric is None:
            metric = ["Precision", "Recall", "MAP", "NDCG", "MRR"]
        elif isinstance(metric, str):
            metric = [metric]
        elif not isinstance(metric, (set, tuple, list)):
            raise TypeError("The type of 'metric' (%s) is invalid!" % metric.__class__.__name__)
        for m in metric:
            if m not in metric_dict:
                raise ValueError("There is not the metric named '%s'!" % metric)
        self.user_pos_train = user_train_dict
        self.user_pos_test = user_test_dict
        self.user_neg_test = user_neg_test
        self.metrics_num = len(metric)
        self.metrics = [metric_dict[m] for m in metric]
        self.num_thread = num_thread
        self.batch_size = batch_size
        self.max_top = top_k if isinstance(top_k, int) else max(top_k)
        # (any top_k, as the reference's evaluate.h:23-50: cut-offs beyond 128 take the one-thread-per-row replay,
        # engine.eval_scores -> nrhip_eval_scores_any_k)
        self.top_show = np.arange(top_k) + 1 if isinstance(top_k, int) else np.sort(top_k)
        self._device_state = None

    def metrics_info(self):
        shown = ["\t".join([("%s@" % re_metric_dict[m] + str(k)).ljust(12) for k in self.top_show])
                 for m in self.metrics]
        return "metrics:\t%s" % "\t".join(shown)

    # ------------------------------------------------------------------ device-side state
    def _device(self, n_items):
        if self._device_state is None or self._device_state["n_items"] != n_items:
            from .... import engine as E
            n_users = 1 + max(max(self.user_pos_train, default=0), max(self.user_pos_test, default=0))
            self._device_state = {
                "n_items": n_items,
                "train": E.DeviceCSR.from_dict(self.user_pos_train, n_users, n_items),
                "test": E.DeviceCSR.from_dict(self.user_pos_test, n_users, n_items),
            }
        return self._device_state

    def _format(self, final_result):
        final_result = np.reshape(final_result, [self.metrics_num, self.max_top])
        final_result = np.reshape(final_result[:, self.top_show - 1], [-1])
        return "\t".join([("%.8f" % x).ljust(12) for x in final_result])

    # ------------------------------------------------------------------ evaluate
    def evaluate(self, model, test_users=None):
        test_users = test_users if test_users is not None else list(self.user_pos_test.keys())
        if not isinstance(test_users, (list, tuple, set, np.ndarray)):
            raise TypeError("'test_user' must be a list, tuple, set or numpy array!")
        test_users = list(test_users)
        if self.user_neg_test is None and hasattr(model, "get_eval_factors"):
            factors = model.get_eval_factors()          # None: this model state has no factor form
            if factors is not None:
                return self._format(self._evaluate_factors(model, test_users, factors))
        return self._format(self._evaluate_scores(model, test_users))

    def _evaluate_factors(self, model, test_users, factors=None):
        import torch
        from .... import parallel
        from ....trainer import FullRankEvaluator
        P, Q = factors if factors is not None else model.get_eval_factors()
        st = self._device(Q.shape[0])
        comm = parallel.get_comm()
        if comm.active:
            return self._evaluate_factors_sharded(comm, model, test_users, P, Q, st)
        if "ranker" not in st:
            st["ranker"] = FullRankEvaluator(st["train"], st["test"], self.metrics, self.max_top,
                                             batch_rows=max(int(self.batch_size), 2048))
        users = torch.tensor(np.asarray(test_users, dtype=np.int32), device=P.device)
        return st["ranker"].evaluate_factors(P, Q, users, exact_mean=True)

    def _evaluate_factors_sharded(self, comm, model, test_users, P, Q, st):
        """One rank of several (SURVEY 8e "Evaluator"): users are independent units, every rank ranks ITS share and the
        per-user metric rows (M·K floats each) are gathered, so that the mean is the float32 np.mean over the same rows
        in the same order as on one GPU (uni_evaluator.py:150-151) — the printed line does not depend on the number of
        ranks.  A model whose user table is row-sharded says which users its rows are (`eval_user_range()` -> (lo, hi):
        P row r = user lo + r; LightGCN's row-sharded engine); otherwise P holds every user and the test users are cut
        into contiguous shares."""
        import torch
        from .... import parallel
        from ....trainer import FullRankEvaluator
        users_all = np.asarray(test_users, dtype=np.int32)
        rng = model.eval_user_range() if hasattr(model, "eval_user_range") else None
        if rng is not None:
            lo, hi = int(rng[0]), min(int(rng[1]), st["train"].n_rows)     # (users past the last one with data have no rows)
            lo = min(lo, hi)
            mine = users_all[(users_all >= lo) & (users_all < hi)]
            key = ("ranker", lo, hi)
            if key not in st:
                st[key] = FullRankEvaluator(st["train"].rows(lo, hi), st["test"].rows(lo, hi), self.metrics, self.max_top,
                                            batch_rows=max(int(self.batch_size), 2048))
            ranker, local = st[key], mine - lo
        else:
            mine = parallel.shard_users(users_all, comm.rank, comm.world)
            if "ranker" not in st:
                st["ranker"] = FullRankEvaluator(st["train"], st["test"], self.metrics, self.max_top,
                                                 batch_rows=max(int(self.batch_size), 2048))
            ranker, local = st["ranker"], mine
        dev = Q.device
        width = self.metrics_num * self.max_top
        if len(mine):
            rows = ranker.evaluate_factors(P, Q, torch.from_numpy(np.ascontiguousarray(local)).to(dev), per_user=True)
            rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32)).to(dev)
        else:
            rows = torch.zeros((0, width), dtype=torch.float32, device=dev)
        got_u, got_r = parallel.gather_rows_by_user(comm, torch.from_numpy(np.ascontiguousarray(mine)).to(dev), rows, dev)
        got_u, got_r = got_u.cpu().numpy(), got_r.cpu().numpy()
        # back into the order of `test_users` (the order the reference's batches run in)
        pos = {int(u): k for k, u in enumerate(got_u)}
        order = np.asarray([pos[int(u)] for u in users_all], dtype=np.int64)
        return np.mean(got_r[order], axis=0)

    def _evaluate_scores(self, model, test_users):
        import torch
        from .... import engine as E
        batch_result = []
        for batch_users in DataIterator(test_users, batch_size=self.batch_size, shuffle=False,
                                        drop_last=False):
            if self.user_neg_test is not None:
                # candidate mode (uni_evaluator.py:123-131): positives first, truth = their slots
                candidate_items = [list(self.user_pos_test[u]) + self.user_neg_test[u]
                                   for u in batch_users]
                test_items = [set(range(len(self.user_pos_test[u]))) for u in batch_users]
                ranking_score = model.predict(batch_users, candidate_items)
                ranking_score = pad_sequences(ranking_score, value=-np.inf, dtype=float_type)
                result = self.eval_score_matrix(ranking_score, test_items, self.metrics,
                                                top_k=self.max_top, thread_num=self.num_thread)
            else:
                ranking_score = model.predict(batch_users, None)
                if isinstance(ranking_score, torch.Tensor):
                    # a device tensor stays where it is; row-strided views (a padded score
                    # slab sliced to [B, n_items]) are consumed in place
                    scores = ranking_score.to(device=E.require_gpu(), dtype=torch.float32)
                    if scores.dim() != 2 or scores.stride(1) != 1:
                        scores = scores.contiguous()
                else:
                    scores = torch.from_numpy(np.array(ranking_score, dtype=float_type)).to(E.require_gpu())
                st = self._device(scores.shape[1])
                users = torch.tensor(np.asarray(batch_users, dtype=np.int32), device=scores.device)
                E.mask_train(scores, users, st["train"])           # -inf on training items
                result = E.eval_scores(scores, st["test"], self.metrics, self.max_top,
                                       users=users).cpu().numpy()
            batch_result.append(result)
        all_user_result = np.concatenate(batch_result, axis=0)
        return np.mean(all_user_result, axis=0)
