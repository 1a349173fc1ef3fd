"""GroupedEvaluator — metrics per user group, groups cut on the number of training
interactions: `group_view=[10,30]` gives (0,10], (10,30]; heavier users are dropped
(evaluator/grouped_evaluator.py:24-112)."""
from collections import OrderedDict

import numpy as np

from ..util.tool import typeassert
from .abstract_evaluator import AbstractEvaluator
from .backend import UniEvaluator


class GroupedEvaluator(AbstractEvaluator):
    @typeassert(user_train_dict=dict, user_test_dict=dict, group_view=list)
    def __init__(self, user_train_dict, user_test_dict, user_neg_test=None, metric=None,
                 group_view=None, top_k=50, batch_size=1024, num_thread=8):
        super(GroupedEvaluator, self).__init__()
        if not isinstance(group_view, list):
            raise TypeError("The type of 'group_view' must be `list`!")
        self.evaluator = UniEvaluator(user_train_dict, user_test_dict, user_neg_test, metric=metric,
                                      top_k=top_k, batch_size=batch_size, num_thread=num_thread)
        self.user_pos_train = user_train_dict
        self.user_pos_test = user_test_dict
        edges = [0] + group_view
        labels = [("(%d,%d]:" % (lo, hi)).ljust(12) for lo, hi in zip(edges[:-1], edges[1:])]
        users = list(self.user_pos_test.keys())
        bucket = np.searchsorted(edges[1:], [len(self.user_pos_train[u]) for u in users])
        self.grouped_user = OrderedDict()
        for g, label in enumerate(labels):
            members = [u for u, b in zip(users, bucket) if b == g]
            if members:
                self.grouped_user[label] = members
        if not self.grouped_user:
            raise ValueError("The splitting of user groups is not suitable!")

    def metrics_info(self):
        return self.evaluator.metrics_info()

    def evaluate(self, model):
        shown = ""
        for group, users in self.grouped_user.items():
            shown = "%s\n%s\t%s" % (shown, group, self.evaluator.evaluate(model, users))
        return shown
