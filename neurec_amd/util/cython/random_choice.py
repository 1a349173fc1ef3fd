"""`randint_choice` / `batch_randint_choice` with the reference's signature and error
behaviour (util/cython/random_choice.pyx:20-89); the draws come from the HIP sampler kernel
(xorshift64*, binary-search exclusion), not from glibc rand()."""
import numpy as np

_state = {"seed": 2018, "calls": 0}


def seed(value):
    """Reseed the device stream (the reference never calls srand; its stream is fixed)."""
    _state["seed"], _state["calls"] = int(value), 0


def _validate(high, size, replace, p, exclusion):
    if size <= 0:
        raise ValueError("'size' must be a positive integer.")
    if not isinstance(replace, bool):
        raise TypeError("'replace' must be bool.")
    if p is not None:
        raise NotImplementedError
    n_excl = len(exclusion) if exclusion is not None else 0
    if exclusion is not None and high <= n_excl:
        raise ValueError("The number of 'exclusion' is greater than 'high'.")
    if replace is False and (high - n_excl <= size):
        raise ValueError("There is not enough integers to be sampled.")


def batch_randint_choice(high, size, replace=True, p=None, exclusion=None):
    """list of per-request samples: an int where size[i]==1, else a list."""
    from ... import engine as E
    if p is not None:
        raise NotImplementedError
    if exclusion is not None and len(size) != len(exclusion):
        raise ValueError("The shape of 'exclusion' is not compatible with the shape of 'size'!")
    sizes = [int(s) for s in size]
    for i, s in enumerate(sizes):
        _validate(high, s, replace, None, None if exclusion is None else exclusion[i])
    if not sizes:
        return []
    ecsr = None
    if exclusion is not None:
        cleaned = [np.unique(np.asarray(list(e), dtype=np.int64)) for e in exclusion]
        cleaned = [e[(e >= 0) & (e < high)].astype(np.int32) for e in cleaned]
        ptr = np.zeros(len(cleaned) + 1, dtype=np.int64)
        ptr[1:] = np.cumsum([len(e) for e in cleaned])
        idx = np.concatenate(cleaned) if ptr[-1] else np.zeros(0, np.int32)
        ecsr = E.DeviceCSR(ptr, idx, high)
    _state["calls"] += 1
    out, off = E.randint_choice_batch(high, sizes, ecsr, replace, _state["seed"], _state["calls"])
    flat = out.cpu().numpy()
    res = []
    for i, s in enumerate(sizes):
        chunk = flat[off[i]:off[i + 1]].tolist()
        res.append(chunk[0] if s == 1 else chunk)
    return res


def randint_choice(high, size=1, replace=True, p=None, exclusion=None):
    _validate(high, size, replace, p, exclusion)
    return batch_randint_choice(high, [size], replace=replace,
                                exclusion=None if exclusion is None else [exclusion])[0]
