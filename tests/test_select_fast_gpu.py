"""Short-row selection (whole row in registers, lane-maxima threshold, one bitonic sort) against the
streaming-ring selection it short-cuts: `nrhip_arg_topk` run in two processes (NEUREC_SELECT_FAST=1 / 0)
on the same matrices — widths around the register capacity, heavy ties, -inf, NaN — must return the
same ranks, and both equal a numpy statement of the order where no ties are involved."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_CHILD = r'''
import json, sys
import numpy as np, torch
sys.path.insert(0, %r)
from neurec_amd import engine as E
out = {}
for cols, k, kind in %r:
    rng = np.random.RandomState(cols * 7 + k)
    rows = 300
    if kind == "float":                     # distinct values per row: no ties
        S = (np.argsort(rng.rand(rows, cols), axis=1).astype(np.float32) - cols // 2) * np.float32(0.37)
    elif kind == "ties":
        S = rng.randint(0, 6, (rows, cols)).astype(np.float32)
    else:                                   # -inf holes (struck items) and a few NaN
        S = rng.randn(rows, cols).astype(np.float32)
        S[rng.rand(rows, cols) < 0.3] = -np.inf
        S[rng.rand(rows, cols) < 0.01] = np.nan
        S[5] = -np.inf
    ld = (cols + 3) // 4 * 4                # 16-byte aligned rows: the float4 instantiation
    buf = np.zeros((rows, ld), np.float32); buf[:, :cols] = S
    d = torch.from_numpy(buf).cuda()
    r = E.arg_topk(d, k, cols=cols)
    out["%%d/%%d/%%s" %% (cols, k, kind)] = r.cpu().numpy().tolist()
print(json.dumps(out))
'''

CASES = [(5, 3, "float"), (63, 20, "float"), (64, 20, "ties"), (65, 21, "holes"), (672, 20, "float"),
         (672, 20, "ties"), (1282, 22, "float"), (1282, 22, "holes"), (2048, 50, "float"), (2049, 20, "float"),
         (300, 63, "float"), (300, 64, "float"), (40, 40, "float"), (1000, 10, "ties")]


def _run(fast):
    env = dict(os.environ, NEUREC_SELECT_FAST=fast)
    out = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, CASES)], env=env, capture_output=True, text=True,
                         timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_register_selection_equals_the_streaming_selection():
    fast, ring = _run("1"), _run("0")
    assert fast.keys() == ring.keys() and len(fast) == len(CASES)
    for key in fast:
        np.testing.assert_array_equal(np.asarray(fast[key]), np.asarray(ring[key]), err_msg=key)
    # tie-free float rows: the order is the descending sort
    for cols, k, kind in CASES:
        if kind != "float":
            continue
        rng = np.random.RandomState(cols * 7 + k)
        S = (np.argsort(rng.rand(300, cols), axis=1).astype(np.float32) - cols // 2) * np.float32(0.37)
        want = np.argsort(-S, axis=1, kind="stable")[:, :min(k, cols)]
        got = np.asarray(fast["%d/%d/%s" % (cols, k, kind)])[:, :min(k, cols)]
        np.testing.assert_array_equal(got, want, err_msg="%d/%d" % (cols, k))
