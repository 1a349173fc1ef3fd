"""Generate tests/golden/*.npz from the REFERENCE's own compiled code (oracle/_ref).

Run in the build container (needs /root/reference to have been compiled by
`make -C oracle ref`):   python tests/golden/make_golden.py

The fixtures are inputs + the outputs the reference produced for them; they are
committed so that the CPU oracle and the HIP engine can be checked against the
reference on boxes where /root/reference does not exist.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
METRICS = [1, 2, 3, 4, 5]


def pack_truth(truth):
    ptr = np.zeros(len(truth) + 1, np.int64)
    ptr[1:] = np.cumsum([len(t) for t in truth])
    return ptr, np.concatenate([np.sort(np.asarray(t, np.int32)) for t in truth]).astype(np.int32)


def make_case(name, scores, truth, top_k, metrics=METRICS):
    scores = np.ascontiguousarray(scores, np.float32)
    res = ref.eval_matrix(scores.copy(), truth, metrics, top_k, threads=4)
    atk = ref.arg_topk(scores.copy(), top_k, threads=4)
    ptr, idx = pack_truth(truth)
    np.savez_compressed(os.path.join(OUT, "eval_%s.npz" % name), scores=scores, truth_ptr=ptr,
                        truth_idx=idx, top_k=np.int32(top_k), metrics=np.asarray(metrics, np.int32),
                        result=res, arg_topk=atk)
    print("eval_%s: scores %s top_k=%d" % (name, scores.shape, top_k))


def main():
    assert ref.available(), "build oracle/_ref first: make -C oracle ref"
    rng = np.random.RandomState(20180917)

    def truth_for(rows, cols, lo=1, hi=40):
        return [np.sort(rng.choice(cols, rng.randint(lo, min(hi, cols) + 1), replace=False)).tolist()
                for _ in range(rows)]

    # SURVEY.md §4 known-answer vector
    kat = np.array([[0.1, 0.9, 0.3, 0.8, 0.2, 0.7, 0.05, 0.6],
                    [0.5, 0.4, 0.3, 0.2, 0.1, 0.0, -0.1, -0.2]], np.float32)
    make_case("kat", kat, [[1, 5, 6], [7]], 4)
    # tie-free gaussian scores, ml-100k-sized rows
    make_case("random", rng.randn(48, 1682), truth_for(48, 1682), 20)
    # heavy ties (quantised scores): exercises the libstdc++ heap order
    make_case("ties", np.round(rng.randn(48, 600) * 2) / 2, truth_for(48, 600), 10)
    # popularity-style rows: identical score vector for every user (Pop.py-like)
    pop = np.tile(np.floor(rng.pareto(1.2, 500)).astype(np.float32), (16, 1))
    make_case("pop", pop, truth_for(16, 500), 20)
    # rows masked with -inf (train items), some rows with fewer than 2K finite scores
    s = rng.randn(32, 300).astype(np.float32)
    for r in range(32):
        n_mask = rng.randint(0, 299) if r % 4 == 0 else rng.randint(0, 60)
        s[r, rng.choice(300, n_mask, replace=False)] = -np.inf
    make_case("masked", s, truth_for(32, 300), 20)
    # truth longer than K, truth of length 1, metric subset order as NeuRec.properties:34
    make_case("long_truth", rng.randn(24, 400), truth_for(24, 400, lo=60, hi=120), 10,
              metrics=[1, 2, 4, 3, 5])
    make_case("single_truth", rng.randn(24, 400), truth_for(24, 400, lo=1, hi=1), 50)
    # fewer columns than 2K (sort_len = cols, evaluate.h:37)
    make_case("narrow", rng.randn(16, 30), truth_for(16, 30, lo=1, hi=5), 20)
    # signed zeros and duplicates at the cut
    z = rng.randn(16, 200).astype(np.float32)
    z[:, ::3] = 0.0
    z[:, 1::7] = -0.0
    make_case("zeros", z, truth_for(16, 200), 15)

    # glibc stream of the reference's Cython sampler (random_choice.pyx), fresh srand(1)
    mod = ref.random_choice_module()
    ctypes.CDLL(None).srand(1)
    first = mod.randint_choice(1000, size=6, replace=True, exclusion=[1, 2, 3])
    excl = [sorted(rng.choice(50, rng.randint(1, 30), replace=False).tolist()) for _ in range(8)]
    sizes = [int(rng.randint(1, 20)) for _ in range(8)]
    batch = mod.batch_randint_choice(50, sizes, replace=True, exclusion=excl)
    norep = mod.randint_choice(40, size=20, replace=False, exclusion=[0, 1, 2, 3, 4])
    np.savez_compressed(
        os.path.join(OUT, "sampler_glibc.npz"), first=np.asarray(first, np.int32),
        sizes=np.asarray(sizes, np.int32),
        excl_ptr=np.cumsum([0] + [len(e) for e in excl]).astype(np.int64),
        excl_idx=np.concatenate(excl).astype(np.int32),
        batch=np.concatenate([np.atleast_1d(b) for b in batch]).astype(np.int32),
        norep=np.asarray(norep, np.int32))
    print("sampler_glibc: first =", first)


if __name__ == "__main__":
    main()
