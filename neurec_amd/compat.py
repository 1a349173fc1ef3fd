"""Import-path compatibility for plugins written against the reference tree.

Reference plugins start with `from model.AbstractRecommender import AbstractRecommender`,
`from util import ...`, `from data import PairwiseSampler`, `from evaluator import
ProxyEvaluator`.  `install()` registers this package's modules under those top-level names so
such files import unchanged (they still need their TensorFlow graph replaced by engine calls).
"""
import importlib
import sys

_ALIASES = {
    "util": "neurec_amd.util",
    "util.tool": "neurec_amd.util.tool",
    "util.logger": "neurec_amd.util.logger",
    "util.configurator": "neurec_amd.util.configurator",
    "util.data_iterator": "neurec_amd.util.data_iterator",
    "util.cython": "neurec_amd.util.cython",
    "util.cython.tools": "neurec_amd.util.cython.tools",
    "util.cython.random_choice": "neurec_amd.util.cython.random_choice",
    "util.cython.arg_topk": "neurec_amd.util.cython.arg_topk",
    "data": "neurec_amd.data",
    "data.sampler": "neurec_amd.data.sampler",
    "data.dataset": "neurec_amd.data.dataset",
    "evaluator": "neurec_amd.evaluator",
    "evaluator.backend": "neurec_amd.evaluator.backend",
    "model": "neurec_amd.model",
    "model.AbstractRecommender": "neurec_amd.model.AbstractRecommender",
}


def install(force=False):
    for alias, target in _ALIASES.items():
        if alias in sys.modules and not force:
            continue
        sys.modules[alias] = importlib.import_module(target)
