"""Samplers with the reference's constructor signatures and iteration contract
(data/sampler.py): iterating yields batches of equal-length fields, `len()` is the number of
batches, the last short batch is kept unless `drop_last`.  Every epoch is formed on the device
— negatives, labels, time-order windows, the epoch permutation and the batching are one launch
(csrc/sampler.hip; trainer.BprEpochSampler for BPR triplets, data/streams.py for the pointwise
and time-ordered kinds) — and batches are Python lists by default, as the reference hands them
to `feed_dict`, or device tensors with `as_tensors=True` (what the HIP models use: the stream
never touches the host).
"""
from .streams import InstanceEpochStream, InstanceRows


class Sampler(object):
    def __len__(self):
        raise NotImplementedError

    def __iter__(self):
        raise NotImplementedError


def _n_batches(n_sample, batch_size, drop_last):
    return n_sample // batch_size if drop_last else -(-n_sample // batch_size)


def _generate_positive_items(user_pos_dict):
    """(user_pos_len, users_list, pos_items_list) of data/sampler.py:24-39, read off the instance index."""
    rows = InstanceRows(user_pos_dict, 0)
    return rows.window_counts(), rows.users().tolist(), rows.positives().tolist()


def _generative_time_order_positive_items(user_pos_dict, high_order=1):
    """(user_pos_len, users_list, recent_items_list, pos_items_list) of data/sampler.py:42-68."""
    if high_order <= 0:
        raise ValueError("'high_order' must be a positive integer.")
    rows = InstanceRows(user_pos_dict, high_order)
    return rows.window_counts(), rows.users().tolist(), rows.recents().tolist(), rows.positives().tolist()


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
        if not self.user_pos_dict:
            raise ValueError("'user_pos_dict' cannot be empty.")
        self._n_sample = sum(len(v) for v in self.user_pos_dict.values())
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
        return _n_batches(self._n_sample, self.batch_size, self.drop_last)


class _StreamSampler(Sampler):
    """Common front end of the three list-built samplers of the reference: an instance index + one device
    stream per epoch (data/streams.py).  `high_order` = 0: plain positives; `pointwise`: labelled instances."""

    def __init__(self, dataset, high_order, by_time, pointwise, neg_num, batch_size, shuffle, drop_last,
                 as_tensors=False, seed=2018):
        if high_order is not None and high_order < 0:
            raise ValueError("'high_order' must be a positive integer.")
        if neg_num <= 0:
            raise ValueError("'neg_num' must be a positive integer.")
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.neg_num, self.item_num, self.as_tensors = neg_num, dataset.num_items, as_tensors
        self.user_pos_dict = dataset.get_user_train_dict(by_time=True) if by_time else dataset.get_user_train_dict()
        if high_order == 0 and by_time:                 # sampler.py:43-44, raised by the window builder
            raise ValueError("'high_order' must be a positive integer.")
        self.rows = InstanceRows(self.user_pos_dict, high_order or 0, self.item_num)
        self.stream = InstanceEpochStream(self.rows, self.item_num, neg_num, pointwise, batch_size, shuffle,
                                          drop_last, seed=seed)

    def __iter__(self):
        return self.stream.batches(as_tensors=self.as_tensors)

    def __len__(self):
        return len(self.stream)

    # the reference's list attributes, formed only if somebody reads them
    @property
    def user_pos_len(self):
        return self.rows.window_counts()

    @property
    def users_list(self):
        return self.rows.users().tolist() * (self.neg_num + 1 if self.stream.pointwise else 1)

    @property
    def pos_items_list(self):
        return self.rows.positives().tolist()

    @property
    def recent_items_list(self):
        return self.rows.recents().tolist() * (self.neg_num + 1 if self.stream.pointwise else 1)

    @property
    def all_labels(self):
        return [1.0] * self.rows.n_inst + [0.0] * (self.rows.n_inst * self.neg_num)


class PointwiseSampler(_StreamSampler):
    """`(users, items, labels)`: every train pair once with label 1 and `neg_num` sampled non-interacted items
    of the same user with label 0 (data/sampler.py:93-155)."""

    def __init__(self, dataset, neg_num=1, batch_size=1024, shuffle=True, drop_last=False, **device):
        super(PointwiseSampler, self).__init__(dataset, None, False, True, neg_num, batch_size, shuffle,
                                               drop_last, **device)


class TimeOrderPointwiseSampler(_StreamSampler):
    """`(users, recent_items, items, labels)`; `recent_items` rows have `high_order` entries when high_order > 1;
    negatives exclude the user's whole train sequence (data/sampler.py:216-289)."""

    def __init__(self, dataset, high_order=1, neg_num=1, batch_size=1024, shuffle=True, drop_last=False, **device):
        super(TimeOrderPointwiseSampler, self).__init__(dataset, high_order, True, True, neg_num, batch_size,
                                                        shuffle, drop_last, **device)


class TimeOrderPairwiseSampler(_StreamSampler):
    """`(users, recent_items, next_items, neg_items)`; `neg_items` rows have `neg_num` entries when neg_num > 1
    (data/sampler.py:292-354)."""

    def __init__(self, dataset, high_order=1, neg_num=1, batch_size=1024, shuffle=True, drop_last=False, **device):
        super(TimeOrderPairwiseSampler, self).__init__(dataset, high_order, True, False, neg_num, batch_size,
                                                       shuffle, drop_last, **device)
