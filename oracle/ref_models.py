"""oracle.ref_models — run the reference's OWN model classes under oracle.tf_shim.

TEST INFRASTRUCTURE ONLY.  Needs the reference tree (NEUREC_REFERENCE, default /root/reference); it
exists in the build container only, so everything made here travels as fixtures under tests/golden/
(tests/golden/make_golden_tfgraph.py).

`load(name)` imports, unchanged and whole, from the reference tree:

    model/general_recommender/<name>.py     the model class (MF / LightGCN / NGCF / MultiVAE)
    model/AbstractRecommender.py            its base class
    util/tool.py, util/learner.py           the graph helpers those files call

with `tensorflow` = oracle.tf_shim and the packages around them replaced by stand-ins that carry no
arithmetic of the path: `util` (re-exports of the real tool/learner functions + an in-memory Logger),
`data` (sampler classes that replay the batches they are given), `evaluator` (a ProxyEvaluator that
records what the model predicts).  Nothing of the reference is copied: the files are executed where
they lie.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

from . import tf_shim

REF = os.environ.get("NEUREC_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "model", "general_recommender", "LightGCN.py"))


# ------------------------------------------------------------------ stand-ins for the surroundings
class MemoryLogger:
    """util.Logger without the file: keeps the lines (`[iter %d : loss : %f, ...`)"""
    lines = []

    def __init__(self, filename=None):
        self.filename = filename

    def info(self, msg):
        MemoryLogger.lines.append(str(msg))

    debug = warning = error = critical = info


class ReplaySampler:
    """data.PairwiseSampler / PointwiseSampler stand-in: iterating yields the batches queued in
    `ReplaySampler.batches` (lists, as the reference's DataIterator hands them over);
    `before_batch(batch)` — when set — is called ahead of every yield (golden scripts fetch the
    pre-update loss terms there)."""
    batches = []
    before_batch = None

    def __init__(self, dataset, neg_num=1, batch_size=1024, shuffle=True, drop_last=False):
        self.args = dict(neg_num=neg_num, batch_size=batch_size, shuffle=shuffle, drop_last=drop_last)

    def __iter__(self):
        for b in ReplaySampler.batches:
            if ReplaySampler.before_batch is not None:
                ReplaySampler.before_batch(b)
            yield tuple(list(map(int, x)) if np.asarray(x).dtype.kind in "iu" else list(map(float, x))
                        for x in b)

    def __len__(self):
        return len(ReplaySampler.batches)


class RecordingEvaluator:
    """evaluator.ProxyEvaluator stand-in: `evaluate(model)` asks the model for the score rows of
    `users` (all training users by default) exactly as UniEvaluator does (`model.predict(batch_users,
    None)`, cpp/uni_evaluator.py:134) and keeps them."""
    users = None
    ratings = []

    def __init__(self, user_train_dict, user_test_dict, user_neg_test=None, metric=None, group_view=None,
                 top_k=50, batch_size=1024, num_thread=8):
        self.train = user_train_dict

    def metrics_info(self):
        return "metrics"

    def evaluate(self, model):
        users = RecordingEvaluator.users
        if users is None:
            users = sorted(self.train.keys())
        r = model.predict(list(users), None)
        RecordingEvaluator.ratings.append(np.array(r, dtype=np.float64 if tf_shim.float_dtype() ==
                                                   tf_shim.torch.float64 else np.float32))
        return "recorded"


class Conf(dict):
    """the two Configurator calls the classes make: conf[key] and conf.params_str()"""

    def params_str(self):
        return "%s_golden" % self["recommender"]


class Dataset:
    """data.dataset.Dataset stand-in built from a train CSR (+ optional test CSR): the five
    accessors the four classes use (dataset.py:261-289)."""

    def __init__(self, train, test=None, name="golden"):
        self.train_matrix = sp.csr_matrix(train)
        self.test_matrix = sp.csr_matrix(test) if test is not None else sp.csr_matrix(train.shape)
        self.num_users, self.num_items = self.train_matrix.shape
        self.dataset_name = name
        self.negative_matrix = None
        self.time_matrix = None

    @staticmethod
    def _dict(m):
        return {u: m.indices[m.indptr[u]:m.indptr[u + 1]].tolist()
                for u in range(m.shape[0]) if m.indptr[u + 1] > m.indptr[u]}

    def get_user_train_dict(self, by_time=False):
        return self._dict(self.train_matrix)

    def get_user_test_dict(self):
        return self._dict(self.test_matrix)

    def get_user_test_neg_dict(self):
        return None

    def get_train_interactions(self):
        coo = self.train_matrix.tocoo()
        return coo.row.tolist(), coo.col.tolist()

    def __str__(self):
        return "golden dataset %dx%d" % (self.num_users, self.num_items)


class _NumpyWithMat:
    """numpy as the reference's pinned 1.16 saw it: `np.mat` (LightGCN.py:153, NGCF.py:363) left
    numpy in 2.0"""
    mat = staticmethod(np.asmatrix)

    def __getattr__(self, attr):
        return getattr(np, attr)


# ------------------------------------------------------------------ loading
def _load_file(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_SHADOWED = ("util", "util.tool", "util.learner", "data", "evaluator", "model", "model.AbstractRecommender",
             "model.general_recommender")


def load(name):
    """the reference module model/general_recommender/<name>.py, executed under the shim"""
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF)
    saved_tf = tf_shim.install()
    saved = {k: sys.modules.get(k) for k in _SHADOWED}
    try:
        tool = _load_file("util.tool", os.path.join(REF, "util", "tool.py"))
        learner = _load_file("util.learner", os.path.join(REF, "util", "learner.py"))
        util = types.ModuleType("util")
        util.__path__ = []
        util.tool, util.learner = tool, learner
        for fn in ("timer", "l2_loss", "inner_product", "log_loss", "csr_to_user_dict", "typeassert",
                   "randint_choice", "pad_sequences", "argmax_top_k"):
            setattr(util, fn, getattr(tool, fn))
        util.Logger = MemoryLogger
        sys.modules["util"] = util

        data = types.ModuleType("data")
        data.PairwiseSampler = data.PointwiseSampler = ReplaySampler
        sys.modules["data"] = data

        ev = types.ModuleType("evaluator")
        ev.ProxyEvaluator = RecordingEvaluator
        sys.modules["evaluator"] = ev

        model_pkg = types.ModuleType("model")
        model_pkg.__path__ = []
        sys.modules["model"] = model_pkg
        _load_file("model.AbstractRecommender", os.path.join(REF, "model", "AbstractRecommender.py"))
        mod = _load_file("model.general_recommender." + name,
                         os.path.join(REF, "model", "general_recommender", name + ".py"))
        sys.modules.pop("model.general_recommender." + name, None)
        if not hasattr(np, "mat"):
            mod.np = _NumpyWithMat()
        return mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        tf_shim.uninstall(saved_tf)


NEUREC_DEFAULTS = {"metric": ["Precision", "Recall", "MAP", "NDCG", "MRR"], "group_view": None, "topk": 20,
                   "test_batch_size": 1024, "num_thread": 8}


def build(name, dataset, hyper, float_width="float32", seed=0):
    """(model, session, module): the reference class `name` constructed on `dataset` with
    NeuRec.properties' evaluator keys + `hyper`, its graph built — main.py:40-44 without TF."""
    tf_shim.set_float(float_width)
    tf_shim.reset_default_graph()
    mod = load(name)
    conf = Conf(NEUREC_DEFAULTS)
    conf["recommender"] = name
    conf.update(hyper)
    sess = tf_shim.Session(seed=seed)
    with np.errstate(divide="ignore"):
        model = getattr(mod, name)(sess, dataset, conf)
    model.build_graph()
    sess.run(tf_shim.global_variables_initializer())
    return model, sess, mod
