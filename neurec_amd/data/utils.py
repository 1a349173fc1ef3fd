"""Host-side loading / filtering / splitting helpers (mirror of data/utils.py).

pandas only; runs once per dataset and is cached on disk by Dataset, exactly as in the
reference.  Not on the accelerated path.
"""
import hashlib
import math
import os

import pandas as pd


def check_md5(file_name):
    if not os.path.isfile(file_name):
        raise FileNotFoundError("There is not file named '%s'!" % file_name)
    digest = hashlib.md5()
    with open(file_name, "rb") as fin:
        digest.update(fin.read())
    return digest.hexdigest()


def load_data(filename, sep, columns):
    return pd.read_csv(filename, sep=sep, header=None, names=columns)


def filter_data(data, user_min=None, item_min=None):
    """Drop NaNs, then items with < item_min and users with < user_min interactions (one pass
    each, items first — data/utils.py:25-38)."""
    data = data.dropna(how="any")
    if item_min is not None and item_min > 0:
        counts = data["item"].map(data["item"].value_counts(sort=False))
        data = data[counts >= item_min]
    if user_min is not None and user_min > 0:
        counts = data["user"].map(data["user"].value_counts(sort=False))
        data = data[counts >= user_min]
    return data


def _ordered(data, by_time):
    return data.sort_values(by=["user", "time"] if by_time else ["user", "item"])


def split_by_ratio(data, ratio=0.8, by_time=True):
    """Per user: first ceil(ratio*n) interactions (time order, or a shuffle) go to train
    (data/utils.py:60-79; the shuffle draws from numpy's global RNG like DataFrame.sample)."""
    head, tail = [], []
    for _, rows in _ordered(data, by_time).groupby(by=["user"]):
        if not by_time:
            rows = rows.sample(frac=1)
        cut = math.ceil(ratio * len(rows))
        head.append(rows.iloc[:cut])
        tail.append(rows.iloc[cut:])
    return pd.concat(head, ignore_index=True), pd.concat(tail, ignore_index=True)


def split_by_loo(data, by_time=True):
    """Leave-one-out: the last interaction of every user with more than 3 goes to test
    (data/utils.py:82-105)."""
    head, tail = [], []
    for _, rows in _ordered(data, by_time).groupby(by=["user"]):
        if len(rows) <= 3:
            head.append(rows)
            continue
        if not by_time:
            rows = rows.sample(frac=1)
        head.append(rows.iloc[:-1])
        tail.append(rows.iloc[-1:])
    return pd.concat(head, ignore_index=True), pd.concat(tail, ignore_index=True)
