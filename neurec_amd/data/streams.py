"""Device-resident instance streams behind the Pointwise / TimeOrder samplers.

The reference builds every epoch of these samplers out of Python lists: a flattened window table
(_generative_time_order_positive_items, data/sampler.py:42-68), per-user rejection sampling in chunks
of 1024 users (_sampling_negative_items, :71-90), list concatenation / repetition for the pointwise
label layout (:131-135, :141-143, :259-266) and a DataIterator that permutes and re-zips everything
(util/data_iterator.py:45-63,133-155).  Here the window table is an index (three prefix arrays, built
once with numpy), and an epoch — windows, negatives, labels, permutation, batching — is ONE launch of
csrc/sampler.hip::sample_instances_kernel; batches are views of its output.
"""
from itertools import chain

import numpy as np
import torch

from .. import engine as E


class InstanceRows:
    """Index of the training instances of {user: item sequence}: row r = the r-th entry of the dict (its
    iteration order is the reference's instance order), instances of row r = its windows of `high_order`
    recent items + the item that follows (high_order = 0: every item is an instance of its own)."""

    def __init__(self, user_pos_dict, high_order=0, n_items=None):
        if not isinstance(user_pos_dict, dict):
            raise TypeError("'user_pos_dict' must be a dict.")
        if not user_pos_dict:
            raise ValueError("'user_pos_dict' cannot be empty.")
        self.high_order = int(high_order)
        R = len(user_pos_dict)
        self.h_row_user = np.fromiter(user_pos_dict.keys(), dtype=np.int64, count=R).astype(np.int32)
        lens = np.fromiter((len(v) for v in user_pos_dict.values()), dtype=np.int64, count=R)
        self.h_seq_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        total = int(self.h_seq_ptr[-1])
        self.h_seq = np.fromiter(chain.from_iterable(user_pos_dict.values()), dtype=np.int64,
                                 count=total).astype(np.int32)
        n_windows = np.maximum(lens - self.high_order, 0)
        self.h_inst_ptr = np.concatenate([[0], np.cumsum(n_windows)]).astype(np.int64)
        self.n_inst = int(self.h_inst_ptr[-1])
        self.h_inst_row = np.repeat(np.arange(R, dtype=np.int32), n_windows)
        # ascending, duplicate-free exclusion set per row (ids outside [0, n_items) can never be drawn)
        row_of_seq = np.repeat(np.arange(R, dtype=np.int64), lens)
        width = int(self.h_seq.max()) + 1 if total else 1
        key = np.unique(row_of_seq * width + self.h_seq)
        e_row, e_item = key // width, key % width
        if n_items is not None:
            keep = e_item < int(n_items)
            e_row, e_item = e_row[keep], e_item[keep]
        self.h_excl = e_item.astype(np.int32)
        self.h_excl_ptr = np.concatenate([[0], np.cumsum(np.bincount(e_row, minlength=R))]).astype(np.int64)
        self._dev = None

    # ---- the reference's list views of the table (built on demand; the device path never needs them)
    def window_counts(self):
        """[[user, #instances]] for rows that have any (user_pos_len of sampler.py:31-35,55-58)"""
        n = np.diff(self.h_inst_ptr)
        return [[int(u), int(c)] for u, c in zip(self.h_row_user, n) if c > 0 or self.high_order == 0]

    def first_positions(self):
        """position in h_seq of every instance's first recent item (= its positive when high_order = 0)"""
        r = self.h_inst_row
        return self.h_seq_ptr[r] + (np.arange(self.n_inst, dtype=np.int64) - self.h_inst_ptr[r])

    def users(self):
        return self.h_row_user[self.h_inst_row]

    def positives(self):
        return self.h_seq[self.first_positions() + self.high_order]

    def recents(self):
        at = self.first_positions()
        if self.high_order == 1:
            return self.h_seq[at]
        return self.h_seq[at[:, None] + np.arange(self.high_order)[None, :]]

    def max_exclusion(self):
        return int(np.diff(self.h_excl_ptr).max()) if len(self.h_excl_ptr) > 1 else 0

    def to_device(self):
        if self._dev is None:
            dev = E.require_gpu()

            def put(a):
                a = np.ascontiguousarray(a)
                return torch.from_numpy(a if a.size else np.zeros(1, a.dtype)).to(dev)
            self.seq_ptr, self.seq = put(self.h_seq_ptr), put(self.h_seq)
            self.excl_ptr, self.excl = put(self.h_excl_ptr), put(self.h_excl)
            self.inst_ptr, self.inst_row = put(self.h_inst_ptr), put(self.h_inst_row)
            self.row_user = put(self.h_row_user)
            self._dev = dev
        return self


class InstanceEpochStream:
    """Epoch streams of one sampler: `pointwise` (user, [recent], item, label) over n_inst·(neg_num+1)
    slots, or pairwise (user, [recent], pos, neg[neg_num]) over n_inst slots.  One launch per epoch."""

    def __init__(self, rows, n_items, neg_num, pointwise, batch_size, shuffle, drop_last, seed=2018):
        if neg_num <= 0:
            raise ValueError("'neg_num' must be a positive integer.")
        self.rows, self.n_items, self.neg_num = rows, int(n_items), int(neg_num)
        self.pointwise, self.batch_size = bool(pointwise), int(batch_size)
        self.shuffle, self.drop_last = bool(shuffle), bool(drop_last)
        self.seed, self.epoch = int(seed), 0
        self.n_slots = rows.n_inst * (self.neg_num + 1 if self.pointwise else 1)

    def __len__(self):
        if self.drop_last:
            return self.n_slots // self.batch_size
        return (self.n_slots + self.batch_size - 1) // self.batch_size

    def _buffers(self):
        """FRESH output buffers for every epoch: the batches handed out are views of them, and the
        reference yields new lists each epoch (data/sampler.py:110-124) — a batch kept past the next __iter__, or a
        second live iterator, must not see the newer epoch's data.  torch's caching allocator makes this a pointer
        bump; the buffers of an epoch are released with its last view."""
        dev = self.rows.to_device()._dev
        n, h = max(self.n_slots, 1), self.rows.high_order
        i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)
        return (i32(n), i32(n * h) if h else None, i32(n),
                None if self.pointwise else i32(n * self.neg_num),
                torch.empty(n, dtype=torch.float32, device=dev) if self.pointwise else None)

    def sample_epoch(self):
        """(users, recent, items, neg, labels) of a fresh epoch — device tensors (None where the kind has none)."""
        if self.n_items <= self.rows.max_exclusion():          # random_choice.pyx:32-33, raised while iterating
            raise ValueError("The number of 'exclusion' is greater than 'high'.")
        out = self._buffers()
        if self.n_slots:
            E.sample_instances_epoch(self.rows, self.n_items, self.neg_num, self.pointwise, self.seed, self.epoch,
                                     self.shuffle, 0, self.n_slots, out)
        self.epoch += 1
        n, h = self.n_slots, self.rows.high_order
        users, recent, items, neg, labels = out
        return (users[:n], None if recent is None else (recent[:n] if h == 1 else recent[:n * h].view(n, h)),
                items[:n], None if neg is None else (neg[:n] if self.neg_num == 1 else
                                                     neg[:n * self.neg_num].view(n, self.neg_num)),
                None if labels is None else labels[:n])

    def batches(self, as_tensors=True):
        """One epoch, batch by batch: tuples of the kind's fields (device views, or Python lists as the reference
        feeds them — one device→host copy per field and epoch, sliced on the host)."""
        fields = [f for f in self.sample_epoch() if f is not None]
        if not as_tensors:
            fields = [f.cpu().numpy() for f in fields]
        B = self.batch_size
        for k in range(len(self)):
            b, e = k * B, min((k + 1) * B, self.n_slots)
            yield tuple(f[b:e] if as_tensors else f[b:e].tolist() for f in fields)
