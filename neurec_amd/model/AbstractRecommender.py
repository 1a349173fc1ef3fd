"""AbstractRecommender — the plugin base class (model/AbstractRecommender.py:23-45).

Subclasses live in `model/<family>_recommender/<Name>.py`, take `(sess, dataset, conf)` (the
first argument was the TensorFlow session; it is accepted and ignored), and implement
`build_graph()`, `train_model()` and `predict(user_ids, candidate_items)`.  The constructor
builds the evaluator and the logger exactly as the reference does, so log paths, the dataset /
config dump and the metric header are unchanged.
"""
import os
import time

from ..evaluator import ProxyEvaluator
from ..util.logger import Logger


def _create_logger(config, data_name):
    """log/<dataset>/<model>/<dataset>_<params[:150]>_<timestamp>.log (AbstractRecommender.py:9-20)."""
    param_str = "%s_%s" % (data_name, config.params_str())
    run_id = "%s_%.8f" % (param_str[:150], time.time())
    log_dir = os.path.join("log", data_name, config["recommender"])
    return Logger(os.path.join(log_dir, run_id + ".log"))


class _Silent(object):
    """the logger of ranks > 0 of a multi-GPU run"""
    def __getattr__(self, name):
        return lambda *a, **k: None


class AbstractRecommender(object):
    def __init__(self, dataset, conf):
        self.evaluator = ProxyEvaluator(dataset.get_user_train_dict(),
                                        dataset.get_user_test_dict(),
                                        dataset.get_user_test_neg_dict(),
                                        metric=conf["metric"],
                                        group_view=conf["group_view"],
                                        top_k=conf["topk"],
                                        batch_size=conf["test_batch_size"],
                                        num_thread=conf["num_thread"])
        # one rank of several (parallel.get_comm): rank 0 writes the run's log, the others only compute
        rank = int(os.environ.get("RANK", "0")) if int(os.environ.get("WORLD_SIZE", "1")) > 1 else 0
        self.logger = _create_logger(conf, dataset.dataset_name) if rank == 0 else _Silent()
        self.logger.info(dataset)
        self.logger.info(conf)

    def build_graph(self):
        raise NotImplementedError

    def train_model(self):
        raise NotImplementedError

    def predict(self, user_ids, items):
        raise NotImplementedError
