"""Samplers with the reference's constructor signatures and iteration contract
(data/sampler.py): iterating yields `(users, pos_items, neg_items)` batches of equal length
<= batch_size, `len()` is the number of batches, the last short batch is kept unless
`drop_last`.  Negative sampling and the epoch shuffle run on the device
(neurec_amd/csrc/sampler.hip); batches are Python lists by default, as the reference hands
them to `feed_dict`, or device tensors with `as_tensors=True` (what the HIP models use, so
that triplets never touch the host).
"""
import numpy as np

from ..util.data_iterator import DataIterator
from ..util.cython.random_choice import batch_randint_choice


class Sampler(object):
    def __len__(self):
        raise NotImplementedError

    def __iter__(self):
        raise NotImplementedError


def _generate_positive_items(user_pos_dict):
    """Flatten {user: items} into user-major parallel lists (data/sampler.py:24-39)."""
    if not isinstance(user_pos_dict, dict):
        raise TypeError("'user_pos_dict' must be a dict.")
    if not user_pos_dict:
        raise ValueError("'user_pos_dict' cannot be empty.")
    users_list, pos_items_list, user_pos_len = [], [], []
    for user, pos_items in user_pos_dict.items():
        user_pos_len.append([user, len(pos_items)])
        users_list.extend([user] * len(pos_items))
        pos_items_list.extend(pos_items)
    return user_pos_len, users_list, pos_items_list


def _sampling_negative_items(user_pos_len, neg_num, item_num, user_pos_dict):
    """`n_u * neg_num` negatives per user, aligned with the positives (data/sampler.py:71-90);
    one device launch for all users instead of 1024-user chunks."""
    if neg_num <= 0:
        raise ValueError("'neg_num' must be a positive integer.")
    users, n_pos = zip(*user_pos_len)
    sizes = [n * neg_num for n in n_pos]
    drawn = batch_randint_choice(item_num, sizes, replace=True,
                                 exclusion=[user_pos_dict[u] for u in users])
    neg_items_list = []
    for neg_items in drawn:
        if isinstance(neg_items, list):
            if neg_num > 1:
                neg_items = np.reshape(neg_items, [-1, neg_num])
            neg_items_list.extend(neg_items)
        else:
            neg_items_list.append(neg_items)
    return neg_items_list


class PairwiseSampler(Sampler):
    """BPR triplets `(user, pos_item, neg_item)`; `neg_items` has shape (batch, neg_num) when
    neg_num > 1 (data/sampler.py:158-213)."""

    def __init__(self, dataset, neg_num=1, batch_size=1024, shuffle=True, drop_last=False,
                 as_tensors=False, seed=2018):
        if neg_num <= 0:
            raise ValueError("'neg_num' must be a positive integer.")
        self.batch_size = batch_size
        self.drop_last = drop_last
        self.shuffle = shuffle
        self.neg_num = neg_num
        self.item_num = dataset.num_items
        self.as_tensors = as_tensors
        self.user_pos_dict = dataset.get_user_train_dict()
        self.user_pos_len, self.users_list, self.pos_items_list = \
            _generate_positive_items(self.user_pos_dict)
        self._seed = seed
        self._device_sampler = None
        # device batches carry their batch plan (ordered row-gradient sums) keyed for this many
        # user rows — the model's table height
        self._plan_users = getattr(dataset, "num_users", None) if as_tensors else None

    def _device(self):
        if self._device_sampler is None:
            from .. import engine as E
            from ..trainer import BprEpochSampler
            csr = E.DeviceCSR.from_dict(self.user_pos_dict, max(self.user_pos_dict) + 1, self.item_num)
            self._device_sampler = BprEpochSampler(csr, self.item_num, neg_num=self.neg_num,
                                                   batch_size=self.batch_size, shuffle=self.shuffle,
                                                   drop_last=self.drop_last, seed=self._seed,
                                                   plan_users=self._plan_users)
        return self._device_sampler

    def epoch_stream(self):
        """One epoch as whole-stream device tensors (users, pos, neg, batch plans or None): the
        batches of __iter__, not cut up — for engines that run an epoch's batch loop natively."""
        return self._device().epoch_stream()

    def __iter__(self):
        for batch in self._device().batches():
            if self.as_tensors:
                yield batch              # a 3-tuple of device tensors; `.plan` = its batch plan
            else:
                users, pos, neg = batch
                yield users.tolist(), pos.tolist(), neg.tolist()

    def __len__(self):
        n_sample = len(self.users_list)
        if self.drop_last:
            return n_sample // self.batch_size
        return (n_sample + self.batch_size - 1) // self.batch_size


class PointwiseSampler(Sampler):
    """`(user, item, label)` with `neg_num` sampled 0-labelled items per positive
    (data/sampler.py:93-155)."""

    def __init__(self, dataset, neg_num=1, batch_size=1024, shuffle=True, drop_last=False):
        if neg_num <= 0:
            raise ValueError("'neg_num' must be a positive integer.")
        self.batch_size = batch_size
        self.drop_last = drop_last
        self.shuffle = shuffle
        self.neg_num = neg_num
        self.item_num = dataset.num_items
        self.user_pos_dict = dataset.get_user_train_dict()
        self.user_pos_len, users_list, self.pos_items_list = \
            _generate_positive_items(self.user_pos_dict)
        self.users_list = users_list * (self.neg_num + 1)
        n_pos = len(self.pos_items_list)
        self.all_labels = [1.0] * n_pos + [0.0] * (n_pos * self.neg_num)

    def __iter__(self):
        neg_items_list = _sampling_negative_items(self.user_pos_len, self.neg_num,
                                                  self.item_num, self.user_pos_dict)
        neg_items = np.reshape(np.array(neg_items_list, dtype=np.int32).T, [-1]).tolist()
        data_iter = DataIterator(self.users_list, self.pos_items_list + neg_items, self.all_labels,
                                 batch_size=self.batch_size, shuffle=self.shuffle,
                                 drop_last=self.drop_last)
        for bat_users, bat_items, bat_labels in data_iter:
            yield bat_users, bat_items, bat_labels

    def __len__(self):
        n_sample = len(self.users_list)
        if self.drop_last:
            return n_sample // self.batch_size
        return (n_sample + self.batch_size - 1) // self.batch_size


def _generative_time_order_positive_items(user_pos_dict, high_order=1):
    """Sliding windows over each user's time-ordered sequence (data/sampler.py:42-68): instance k
    of a user pairs the `high_order` items starting at position k with the item that follows."""
    if high_order <= 0:
        raise ValueError("'high_order' must be a positive integer.")
    if not isinstance(user_pos_dict, dict):
        raise TypeError("'user_pos_dict' must be a dict.")
    if not user_pos_dict:
        raise ValueError("'user_pos_dict' cannot be empty.")
    users_list, recent_items_list, pos_items_list, user_pos_len = [], [], [], []
    for user, seq_items in user_pos_dict.items():
        num_instance = len(seq_items) - high_order
        if num_instance <= 0:
            continue
        user_pos_len.append([user, num_instance])
        users_list.extend([user] * num_instance)
        if high_order == 1:
            recent_items_list.extend(seq_items[:num_instance])
        else:
            recent_items_list.extend([list(seq_items[k:k + high_order]) for k in range(num_instance)])
        pos_items_list.extend(seq_items[high_order:])
    return user_pos_len, users_list, recent_items_list, pos_items_list


class _TimeOrderBase(Sampler):
    def __init__(self, dataset, high_order, neg_num, batch_size, shuffle, drop_last):
        if high_order < 0:
            raise ValueError("'high_order' must be a positive integer.")
        if neg_num <= 0:
            raise ValueError("'neg_num' must be a positive integer.")
        self.batch_size = batch_size
        self.drop_last = drop_last
        self.shuffle = shuffle
        self.neg_num = neg_num
        self.item_num = dataset.num_items
        self.user_pos_dict = dataset.get_user_train_dict(by_time=True)

    def __len__(self):
        n_sample = len(self.users_list)
        if self.drop_last:
            return n_sample // self.batch_size
        return (n_sample + self.batch_size - 1) // self.batch_size


class TimeOrderPointwiseSampler(_TimeOrderBase):
    """`(user, recent_items, item, label)`; negatives exclude the user's whole train sequence and
    are drawn on the device (data/sampler.py:216-289)."""

    def __init__(self, dataset, high_order=1, neg_num=1, batch_size=1024, shuffle=True, drop_last=False):
        super(TimeOrderPointwiseSampler, self).__init__(dataset, high_order, neg_num, batch_size,
                                                        shuffle, drop_last)
        self.user_pos_len, users_list, recent_items_list, self.pos_items_list = \
            _generative_time_order_positive_items(self.user_pos_dict, high_order=high_order)
        self.users_list = users_list * (self.neg_num + 1)
        self.recent_items_list = recent_items_list * (self.neg_num + 1)
        n_pos = len(self.pos_items_list)
        self.all_labels = [1.0] * n_pos + [0.0] * (n_pos * self.neg_num)

    def __iter__(self):
        neg_items_list = _sampling_negative_items(self.user_pos_len, self.neg_num,
                                                  self.item_num, self.user_pos_dict)
        neg_items = np.reshape(np.array(neg_items_list, dtype=np.int32).T, [-1]).tolist()
        data_iter = DataIterator(self.users_list, self.recent_items_list,
                                 self.pos_items_list + neg_items, self.all_labels,
                                 batch_size=self.batch_size, shuffle=self.shuffle,
                                 drop_last=self.drop_last)
        for bat_users, bat_recent, bat_next, bat_labels in data_iter:
            yield bat_users, bat_recent, bat_next, bat_labels


class TimeOrderPairwiseSampler(_TimeOrderBase):
    """`(user, recent_items, next_item, neg_items)` (data/sampler.py:292-354)."""

    def __init__(self, dataset, high_order=1, neg_num=1, batch_size=1024, shuffle=True, drop_last=False):
        super(TimeOrderPairwiseSampler, self).__init__(dataset, high_order, neg_num, batch_size,
                                                       shuffle, drop_last)
        self.user_pos_len, self.users_list, self.recent_items_list, self.pos_items_list = \
            _generative_time_order_positive_items(self.user_pos_dict, high_order=high_order)

    def __iter__(self):
        neg_items_list = _sampling_negative_items(self.user_pos_len, self.neg_num,
                                                  self.item_num, self.user_pos_dict)
        data_iter = DataIterator(self.users_list, self.recent_items_list, self.pos_items_list,
                                 neg_items_list, batch_size=self.batch_size, shuffle=self.shuffle,
                                 drop_last=self.drop_last)
        for bat_users, bat_recent, bat_pos, bat_neg in data_iter:
            yield bat_users, bat_recent, bat_pos, bat_neg
