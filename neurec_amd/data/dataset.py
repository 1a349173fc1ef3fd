"""Dataset — loads `<name>.rating` (or `.train/.test`), filters, splits, remaps ids, caches the
split under `_tmp_<name>/` keyed by the md5 of the source files, and exposes the CSR matrices
and dict views the hot path consumes.

Behavioural mirror of data/dataset.py:18-289 (file formats, cache file names, id remapping by
first appearance, `num_users = max id + 1`, the info string).  One extension: when the data
directory is read-only the cache goes to `data.cache.path` (default: a per-user temp dir)
instead of next to the data.
"""
import os
import tempfile

import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix

from ..util.logger import Logger
from ..util.tool import csr_to_user_dict, csr_to_user_dict_bytime, randint_choice
from .utils import check_md5, filter_data, split_by_loo, split_by_ratio

_FORMATS = {"UIRT": ["user", "item", "rating", "time"], "UIR": ["user", "item", "rating"],
            "UI": ["user", "item"], "UIT": ["user", "item", "time"]}


class Dataset(object):
    def __init__(self, conf):
        self.train_matrix = None
        self.test_matrix = None
        self.time_matrix = None
        self.negative_matrix = None
        self.userids = None
        self.itemids = None
        self.num_users = None
        self.num_items = None
        self.dataset_name = conf["data.input.dataset"]
        self._load_data(conf)

    # ------------------------------------------------------------------ paths and cache
    def _get_data_path(self, config):
        data_path = config["data.input.path"]
        ori_prefix = os.path.join(data_path, self.dataset_name)
        cache_root = data_path
        if "data.cache.path" in config:
            cache_root = config["data.cache.path"]
        elif not os.access(data_path, os.W_OK):
            cache_root = os.path.join(tempfile.gettempdir(), "neurec_amd_cache_%d" % os.getuid())
        saved_path = os.path.join(cache_root, "_tmp_" + self.dataset_name)
        saved_prefix = "%s_%s_u%d_i%d" % (self.dataset_name, config["splitter"],
                                          config["user_min"], config["item_min"])
        if "by_time" in config and config["by_time"] is True:
            saved_prefix += "_by_time"
        return ori_prefix, os.path.join(saved_path, saved_prefix)

    @staticmethod
    def _source_md5(splitter, ori_prefix):
        if splitter in ("loo", "ratio"):
            return [check_md5(ori_prefix + ".rating")]
        if splitter == "given":
            return [check_md5(ori_prefix + ".train"), check_md5(ori_prefix + ".test")]
        raise ValueError("'%s' is an invalid splitter!" % splitter)

    def _check_saved_data(self, splitter, ori_prefix, saved_prefix):
        want = self._source_md5(splitter, ori_prefix)
        ok = False
        if os.path.isfile(saved_prefix + ".md5"):
            with open(saved_prefix + ".md5") as fin:
                ok = [line.strip() for line in fin.readlines()] == want
        for postfix in (".train", ".test", ".user2id", ".item2id"):
            ok = ok and os.path.isfile(saved_prefix + postfix)
        return ok

    # ------------------------------------------------------------------ load
    def _load_data(self, config):
        file_format = config["data.column.format"]
        if file_format not in _FORMATS:
            raise ValueError("'%s' is an invalid data column format!" % file_format)
        columns = _FORMATS[file_format]
        ori_prefix, saved_prefix = self._get_data_path(config)
        splitter = config["splitter"]
        sep = config["data.convert.separator"]

        if self._check_saved_data(splitter, ori_prefix, saved_prefix):
            print("load saved data...")
            train_data = pd.read_csv(saved_prefix + ".train", sep=sep, header=None, names=columns)
            test_data = pd.read_csv(saved_prefix + ".test", sep=sep, header=None, names=columns)
            user_map = pd.read_csv(saved_prefix + ".user2id", sep=sep, header=None, names=["user", "id"])
            item_map = pd.read_csv(saved_prefix + ".item2id", sep=sep, header=None, names=["item", "id"])
            self.userids = dict(zip(user_map["user"], user_map["id"]))
            self.itemids = dict(zip(item_map["item"], item_map["id"]))
        else:
            print("split and save data...")
            by_time = config["by_time"] if file_format in {"UIRT", "UIT"} else False
            train_data, test_data = self._split_data(ori_prefix, saved_prefix, columns, by_time, config)

        all_data = pd.concat([train_data, test_data])
        self.num_users = int(max(all_data["user"])) + 1
        self.num_items = int(max(all_data["item"])) + 1
        self.num_ratings = len(all_data)
        shape = (self.num_users, self.num_items)

        if file_format in {"UI", "UIT"}:
            train_ratings, test_ratings = [1.0] * len(train_data), [1.0] * len(test_data)
        else:
            train_ratings, test_ratings = train_data["rating"], test_data["rating"]
        self.train_matrix = csr_matrix((train_ratings, (train_data["user"], train_data["item"])), shape=shape)
        self.test_matrix = csr_matrix((test_ratings, (test_data["user"], test_data["item"])), shape=shape)
        if file_format in {"UIRT", "UIT"}:
            self.time_matrix = csr_matrix((train_data["time"], (train_data["user"], train_data["item"])),
                                          shape=shape)
        self.negative_matrix = self._load_test_neg_items(all_data, config, saved_prefix)

    def _split_data(self, ori_prefix, saved_prefix, columns, by_time, config):
        splitter = config["splitter"]
        sep = config["data.convert.separator"]
        os.makedirs(os.path.dirname(saved_prefix), exist_ok=True)
        if splitter in ("loo", "ratio"):
            rating_file = ori_prefix + ".rating"
            all_data = pd.read_csv(rating_file, sep=sep, header=None, names=columns)
            kept = filter_data(all_data, user_min=config["user_min"], item_min=config["item_min"])
            if splitter == "ratio":
                train_data, test_data = split_by_ratio(kept, ratio=config["ratio"], by_time=by_time)
            else:
                train_data, test_data = split_by_loo(kept, by_time=by_time)
        elif splitter == "given":
            train_data = pd.read_csv(ori_prefix + ".train", sep=sep, header=None, names=columns)
            test_data = pd.read_csv(ori_prefix + ".test", sep=sep, header=None, names=columns)
        else:
            raise ValueError("'%s' is an invalid splitter!" % splitter)
        with open(saved_prefix + ".md5", "w") as md5_out:
            md5_out.write("\n".join(self._source_md5(splitter, ori_prefix)))

        # ids are assigned in order of first appearance in train ++ test (dataset.py:167-176)
        both = pd.concat([train_data, test_data])
        self.userids = {u: k for k, u in enumerate(both["user"].unique())}
        self.itemids = {i: k for k, i in enumerate(both["item"].unique())}
        train_data, test_data = train_data.copy(), test_data.copy()
        for frame in (train_data, test_data):
            frame["user"] = frame["user"].map(self.userids)
            frame["item"] = frame["item"].map(self.itemids)

        np.savetxt(saved_prefix + ".train", train_data, fmt="%d", delimiter=sep)
        np.savetxt(saved_prefix + ".test", test_data, fmt="%d", delimiter=sep)
        np.savetxt(saved_prefix + ".user2id", [[u, k] for u, k in self.userids.items()], fmt="%s", delimiter=sep)
        np.savetxt(saved_prefix + ".item2id", [[i, k] for i, k in self.itemids.items()], fmt="%s", delimiter=sep)

        neg_item_file = ori_prefix + ".neg"
        if os.path.isfile(neg_item_file):                 # remap provided test negatives
            rows = []
            with open(neg_item_file) as fin:
                for line in fin.readlines():
                    tokens = line.strip().split(sep)
                    rows.append([self.userids[tokens[0]]] + [self.itemids[i] for i in tokens[1:]])
            np.savetxt("%s.neg%d" % (saved_prefix, len(rows[0]) - 1), rows, fmt="%d", delimiter=sep)

        remapped = pd.concat([train_data, test_data])
        self.num_users = int(max(remapped["user"])) + 1
        self.num_items = int(max(remapped["item"])) + 1
        self.num_ratings = len(remapped)
        logger = Logger(saved_prefix + ".info")
        logger.info(os.path.basename(saved_prefix))
        logger.info(self.__str__())
        return train_data, test_data

    def _load_test_neg_items(self, all_data, config, saved_prefix):
        """`rec.evaluate.neg` sampled (or loaded) negatives per user (dataset.py:212-242)."""
        number_neg = config["rec.evaluate.neg"]
        if not number_neg or number_neg <= 0:
            return None
        sep = config["data.convert.separator"]
        neg_items_file = "%s.neg%d" % (saved_prefix, number_neg)
        if os.path.isfile(neg_items_file):
            neg_items = pd.read_csv(neg_items_file, sep=sep, header=None)
        else:
            lines = []
            for user, u_data in all_data.groupby(["user"]):
                user = user[0] if isinstance(user, tuple) else user
                picks = randint_choice(self.num_items, size=number_neg, replace=False,
                                       exclusion=u_data["item"].tolist())
                lines.append([user] + list(picks))
            neg_items = pd.DataFrame(lines)
            np.savetxt(neg_items_file, neg_items, fmt="%d", delimiter=sep)
        user_list, item_list = [], []
        for line in neg_items.values:
            user_list.extend([line[0]] * (len(line) - 1))
            item_list.extend(line[1:])
        return csr_matrix(([1] * len(user_list), (user_list, item_list)),
                          shape=(self.num_users, self.num_items))

    # ------------------------------------------------------------------ views
    def __str__(self):
        nu, ni, nr = self.num_users, self.num_items, self.num_ratings
        sparsity = 1 - 1.0 * nr / (nu * ni)
        return "\n".join(["Dataset name: %s" % self.dataset_name,
                          "The number of users: %d" % nu,
                          "The number of items: %d" % ni,
                          "The number of ratings: %d" % nr,
                          "Average actions of users: %.2f" % (1.0 * nr / nu),
                          "Average actions of items: %.2f" % (1.0 * nr / ni),
                          "The sparsity of the dataset: %.6f%%" % (sparsity * 100)])

    __repr__ = __str__

    def get_user_train_dict(self, by_time=False):
        if by_time:
            return csr_to_user_dict_bytime(self.time_matrix, self.train_matrix)
        return csr_to_user_dict(self.train_matrix)

    def get_user_test_dict(self):
        return csr_to_user_dict(self.test_matrix)

    def get_user_test_neg_dict(self):
        if self.negative_matrix is None:
            return None
        return csr_to_user_dict(self.negative_matrix)

    def get_train_interactions(self):
        """(users, items) of every training interaction, row-major."""
        coo = self.train_matrix.tocoo()
        order = np.lexsort((coo.col, coo.row))
        return coo.row[order].tolist(), coo.col[order].tolist()

    def to_csr_matrix(self):
        return self.train_matrix.copy()
