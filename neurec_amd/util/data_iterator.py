"""DataIterator — shuffler/batcher over parallel Python sequences.

Semantics of util/data_iterator.py:45-210: one `np.random.permutation(n)` per pass when
`shuffle` (numpy's global RNG, seeded by main.py:10), batches of `batch_size` consecutive
(permuted) positions, the last short batch kept unless `drop_last`, each batch returned as
one list per data sequence (a flat list when a single sequence was given).
"""
import numpy as np


class DataIterator(object):
    def __init__(self, *data, batch_size=1, shuffle=False, drop_last=False):
        data = list(data)
        for seq in data:
            if len(seq) != len(data[0]):
                raise ValueError("The length of the given data are not equal!")
        if not isinstance(batch_size, int) or isinstance(batch_size, bool) or batch_size <= 0:
            raise ValueError("batch_size should be a positive integeral value, "
                             "but got batch_size={}".format(batch_size))
        if not isinstance(drop_last, bool):
            raise ValueError("drop_last should be a boolean value, but got "
                             "drop_last={}".format(drop_last))
        self.data = data
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.drop_last = drop_last

    def _n(self):
        return len(self.data[0]) if self.data else 0

    def __len__(self):
        n = self._n()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self._n()
        order = np.random.permutation(n).tolist() if self.shuffle else range(n)
        order = list(order)
        single = len(self.data) == 1
        for k in range(len(self)):
            picks = order[k * self.batch_size:(k + 1) * self.batch_size]
            columns = [[seq[i] for i in picks] for seq in self.data]
            yield columns[0] if single else columns
