"""kernel table of an evaluation with 30 rows redone (scripts/prof_py.sh redo scripts/exp_eval_redo_prof.py)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import FullRankEvaluator

train, test = synth.interactions_around_test(synth.load_test_split(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "gowalla_test_split.npz")), 810128, seed=2018)
U, I = train.shape
rng = np.random.RandomState(0)
P = torch.from_numpy(synth.xavier_uniform(U, 64, rng)).cuda()
Q = torch.from_numpy(synth.xavier_uniform(I, 64, rng)).cuda()
trc, tec = E.DeviceCSR.from_scipy(train), E.DeviceCSR.from_scipy(test)
users = torch.from_numpy(np.flatnonzero(np.diff(test.indptr) > 0).astype(np.int32)).cuda()
n = users.numel()
ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=32768)
ev.int8_retry = 0
marked = torch.from_numpy(np.random.RandomState(1).choice(n, 30, replace=False)).cuda()
read = ev._read_native_sums
state = {"first": True}


def patched():
    both = read()
    if state["first"]:
        ev._flags_buf[marked] = 2
        both[-2] += 30; both[-1] += 30
    state["first"] = not state["first"]
    return both


ev._read_native_sums = patched
for _ in range(12):
    ev.evaluate_factors(P, Q, users)
torch.cuda.synchronize()
print("rows redone:", ev.n_flagged)
