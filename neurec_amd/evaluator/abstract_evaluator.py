"""Base class of all evaluators (evaluator/abstract_evaluator.py)."""


class AbstractEvaluator(object):
    def __init__(self):
        pass

    def metrics_info(self):
        """Header string such as "Precision@10    Precision@20    NDCG@10    NDCG@20"."""
        raise NotImplementedError

    def evaluate(self, model):
        """One-line result string such as "0.18663847    0.11239596 ..."."""
        raise NotImplementedError
