"""Host-side helpers with the reference's names and behaviour (util/tool.py).

The TensorFlow graph builders of the reference (`inner_product`, `l2_loss`, `log_loss`,
`get_initializer`, `activation_function`) are provided as array functions: the engine has
no graph, the math of the hot path lives in the HIP kernels, and these exist so that
plugin code written against `util.tool` keeps importing and computing the same values.
"""
import heapq
import itertools
import time
from functools import wraps
from inspect import signature

import numpy as np


def get_data_format(data_format):
    table = {"UIRT": ["user", "item", "rating", "time"], "UIR": ["user", "item", "rating"],
             "UIT": ["user", "item", "time"], "UI": ["user", "item"]}
    if data_format not in table:
        raise ValueError("please choose a correct data format. ")
    return table[data_format]


def csr_to_user_dict(train_matrix):
    """{row: [column ids ascending]} for non-empty rows (util/tool.py:56-65)."""
    m = train_matrix.tocsr()
    if not m.has_sorted_indices:
        m = m.copy()
        m.sort_indices()
    out = {}
    indptr, indices = m.indptr, m.indices
    for row in range(m.shape[0]):
        b, e = indptr[row], indptr[row + 1]
        if e > b:
            out[row] = indices[b:e].tolist()
    return out


def csr_to_user_dict_bytime(time_matrix, train_matrix):
    """Same, each user's items ordered by interaction time (util/tool.py:68-76)."""
    out = {}
    for u, items in csr_to_user_dict(train_matrix).items():
        out[u] = np.array(sorted(items, key=lambda x: time_matrix[u, x]), dtype=np.int32).tolist()
    return out


def get_initializer(init_method, stddev, seed=None):
    """Returns `init(shape) -> float32 ndarray` with the distribution TF would use for
    `init_method` (util/tool.py:79-97); the random stream itself is numpy's, not Philox."""
    rng = np.random.RandomState(seed)

    def fans(shape):
        return (shape[0], shape[1]) if len(shape) >= 2 else (shape[0], shape[0])

    def tnormal(shape, sd):
        out = rng.normal(0.0, sd, size=shape)
        bad = np.abs(out) > 2 * sd
        while bad.any():
            out[bad] = rng.normal(0.0, sd, size=int(bad.sum()))
            bad = np.abs(out) > 2 * sd
        return out

    def build(shape):
        shape = tuple(shape)
        fi, fo = fans(shape)
        if init_method == "uniform":
            w = rng.uniform(-stddev, stddev, size=shape)
        elif init_method == "normal":
            w = rng.normal(0.0, stddev, size=shape)
        elif init_method == "xavier_normal":
            # tf.contrib.layers.xavier_initializer(uniform=False) = variance_scaling(factor=1,
            # FAN_AVG): truncated normal with stddev sqrt(1.3 * factor / n), n = (fan_in + fan_out) / 2
            w = tnormal(shape, np.sqrt(1.3 * 2.0 / (fi + fo)))
        elif init_method == "xavier_uniform":
            lim = np.sqrt(6.0 / (fi + fo))
            w = rng.uniform(-lim, lim, size=shape)
        elif init_method == "he_normal":
            w = tnormal(shape, np.sqrt(1.3 * 2.0 / fi))
        elif init_method == "he_uniform":
            lim = np.sqrt(3.0 * 2.0 / fi)
            w = rng.uniform(-lim, lim, size=shape)
        else:                                   # 'tnormal' and anything unknown
            w = tnormal(shape, stddev)
        return w.astype(np.float32)
    return build


def randint_choice(high, size=None, replace=True, p=None, exclusion=None):
    """numpy sampler of util/tool.py:116-129 (used off the hot path by a few models)."""
    a = np.arange(high)
    if exclusion is not None:
        p = np.ones_like(a) if p is None else np.array(p, copy=True)
        p = p.flatten().astype(np.float64)
        p[exclusion] = 0
        p = p / np.sum(p)
    return np.random.choice(a, size=size, replace=replace, p=p)


def typeassert(*type_args, **type_kwargs):
    def decorate(func):
        sig = signature(func)
        bound_types = sig.bind_partial(*type_args, **type_kwargs).arguments

        @wraps(func)
        def wrapper(*args, **kwargs):
            for name, value in sig.bind(*args, **kwargs).arguments.items():
                if name in bound_types and not isinstance(value, bound_types[name]):
                    raise TypeError("Argument {} must be {}".format(name, bound_types[name]))
            return func(*args, **kwargs)
        return wrapper
    return decorate


def argmax_top_k(a, top_k=50):
    best = heapq.nlargest(top_k, zip(a, itertools.count()))
    return np.array([idx for _, idx in best], dtype=np.intc)


def pad_sequences(sequences, value=0., max_len=None, padding="post", truncating="post",
                  dtype=np.int32):
    """Pad/truncate a list of sequences to one length (util/tool.py:154-195)."""
    if max_len is None:
        max_len = int(np.max([len(x) for x in sequences]))
    out = np.full([len(sequences), max_len], value, dtype=dtype)
    for row, seq in enumerate(sequences):
        if not len(seq):
            continue
        if truncating == "pre":
            cut = seq[-max_len:]
        elif truncating == "post":
            cut = seq[:max_len]
        else:
            raise ValueError('Truncating type "%s" not understood' % truncating)
        if padding == "post":
            out[row, :len(cut)] = cut
        elif padding == "pre":
            out[row, -len(cut):] = cut
        else:
            raise ValueError('Padding type "%s" not understood' % padding)
    return out


def inner_product(a, b, name="inner_product"):
    return (a * b).sum(-1)


def l2_loss(*params):
    return sum((w * w).sum() / 2 for w in params)


def log_loss(yij, name="log_loss"):
    """BPR loss -log(sigmoid(y)) elementwise."""
    y = np.asarray(yij)
    return np.logaddexp(0.0, -y)


def timer(func):
    @wraps(func)
    def wrapper(*args, **kwargs):
        start = time.time()
        result = func(*args, **kwargs)
        print("%s function cost: %fs" % (func.__name__, time.time() - start))
        return result
    return wrapper
