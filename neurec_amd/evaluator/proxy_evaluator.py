"""ProxyEvaluator — the evaluator facade models talk to (evaluator/proxy_evaluator.py:41-108)."""
from ..util.tool import typeassert
from .abstract_evaluator import AbstractEvaluator
from .backend import UniEvaluator
from .grouped_evaluator import GroupedEvaluator


class ProxyEvaluator(AbstractEvaluator):
    @typeassert(user_train_dict=dict, user_test_dict=dict)
    def __init__(self, user_train_dict, user_test_dict, user_neg_test=None, metric=None,
                 group_view=None, top_k=50, batch_size=1024, num_thread=8):
        super(ProxyEvaluator, self).__init__()
        common = dict(metric=metric, top_k=top_k, batch_size=batch_size, num_thread=num_thread)
        if group_view is not None:
            self.evaluator = GroupedEvaluator(user_train_dict, user_test_dict, user_neg_test,
                                              group_view=group_view, **common)
        else:
            self.evaluator = UniEvaluator(user_train_dict, user_test_dict, user_neg_test, **common)

    def metrics_info(self):
        return self.evaluator.metrics_info()

    def evaluate(self, model):
        return self.evaluator.evaluate(model)
