"""The drop-in entry point as one rank of several (north_star: "existing configs drop in ... tables row-shard across
the GPUs"): `python -m torch.distributed.run --nproc-per-node 2 -m neurec_amd.main --recommender=LightGCN ...` in a
directory with NeuRec.properties + conf/ + a .rating file, two ranks sharing the one visible GPU over gloo, against
the same command line on one process.  `batch_size` stays the GLOBAL batch, so the N-rank run IS the one-GPU run,
partitioned:
  * --dp_mode=rowshard (tables row-sharded, all-to-all lookups, a rank evaluates ITS users against the gathered item
    table): every epoch's metric line identical to the one-GPU log, character for character;
  * --dp_mode=colshard (the default: every rank holds embed_size / N columns): the inner products are sums of per-rank
    partial dots, so the lines agree to north_star's 1e-5.
Rank 0 writes the one log; the per-user metric rows are gathered so that the mean is the reference's float32 mean."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from neurec_amd import defaults
from test_dropin_gpu import _write_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
ARGS = ["--recommender=LightGCN", "--epochs=4", "--batch_size=256", "--embed_size=64", "--n_layers=2", "--lr=0.01"]
MF_ARGS = ["--recommender=MF", "--epochs=6", "--batch_size=128", "--learning_rate=0.01", "--reg_mf=0.001", "--verbose=2"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(folder, ranks, extra, args=None, model="LightGCN", raw=False):
    args = ARGS if args is None else args
    os.makedirs(folder, exist_ok=True)
    _write_dataset(folder, n_users=230, n_items=150)
    defaults.write_default_configs(folder, overrides={"data.input.path": os.path.join(folder, "dataset"),
                                                      "data.input.dataset": "toy", "test_batch_size": "64"})
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), NEUREC_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if ranks == 1:
        cmd = [sys.executable, "-m", "neurec_amd.main"] + args + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m", "neurec_amd.main"] + args + extra
    out = subprocess.run(cmd, cwd=folder, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    logs = os.path.join(folder, "log", "toy", model)
    files = os.listdir(logs)
    assert len(files) == 1, files                                  # rank 0 writes the run's one log
    with open(os.path.join(logs, files[0])) as f:
        text = f.read()
    return text if raw else re.findall(r"epoch (\d+):\t(.+)", text)


def test_two_rank_lightgcn_drop_in_equals_the_one_gpu_run(tmp_path):
    one = _run(str(tmp_path / "one"), 1, [])
    rows = _run(str(tmp_path / "rows"), 2, ["--dp_mode=rowshard"])
    cols = _run(str(tmp_path / "cols"), 2, [])
    assert [e[0] for e in one] == [e[0] for e in rows] == [e[0] for e in cols] == ["0", "1", "2", "3"]
    for a, b in zip(one, rows):
        assert a[1] == b[1], (a, b)                                # row-sharded: the same line, to the last digit
    va = np.asarray([[float(x) for x in e[1].split("\t")] for e in one])
    vc = np.asarray([[float(x) for x in e[1].split("\t")] for e in cols])
    assert np.abs(va - vc).max() <= 1e-5 and va[-1].max() > 0.05   # column-sharded: 1e-5; and the model has learnt


def test_two_rank_mf_drop_in_equals_the_one_gpu_run(tmp_path):
    """conf/MF.properties as shipped (BPR, adam) on two ranks: both tables row-sharded with their Adam moments
    (sharded.ShardedMF — north_star's "row-shard ... with all-to-all for cross-shard lookups"), a rank evaluates ITS
    users against the gathered item table: the metric lines of the log are the one-GPU run's character for character
    (the tables are bit-identical), the logged epoch losses agree to fp32 rounding (two per-rank sums added)."""
    one = _run(str(tmp_path / "one"), 1, [], MF_ARGS, "MF", raw=True)
    two = _run(str(tmp_path / "two"), 2, [], MF_ARGS, "MF", raw=True)
    ev = lambda t: re.findall(r"epoch (\d+):\t(.+)", t)
    assert [e[0] for e in ev(one)] == ["2", "4", "6"] and ev(one) == ev(two)
    lo = lambda t: [float(x) for x in re.findall(r"\[iter \d+ : loss : ([0-9.]+),", t)]
    assert len(lo(one)) == 6 and np.allclose(lo(one), lo(two), rtol=1e-5, atol=0)
    assert lo(one)[-1] < lo(one)[0]


def test_two_rank_ngcf_drop_in_tracks_the_one_gpu_run(tmp_path):
    """conf/NGCF.properties (ngcf, adam) on two ranks: node rows sharded, the layer weights replicated and their
    gradients all-reduced (sharded_ngcf.ShardedNGCF).  With message dropout off the run is deterministic and differs
    from one GPU only by the association of the weight-gradient sums: metric lines and epoch losses agree to 1e-5."""
    args = ["--recommender=NGCF", "--epochs=4", "--batch_size=128", "--learning_rate=0.005", "--mess_dropout_ratio=0.0",
            "--verbose=2"]
    one = _run(str(tmp_path / "one"), 1, [], args, "NGCF", raw=True)
    two = _run(str(tmp_path / "two"), 2, [], args, "NGCF", raw=True)
    ev = lambda t: np.asarray([[float(x) for x in e[1].split("\t")] for e in re.findall(r"epoch (\d+):\t(.+)", t)])
    lo = lambda t: [float(x) for x in re.findall(r"\[iter \d+ : loss : ([0-9.]+),", t)]
    assert ev(one).shape == (2, 10) and np.abs(ev(one) - ev(two)).max() <= 1e-5
    assert len(lo(one)) == 4 and np.allclose(lo(one), lo(two), rtol=1e-5, atol=0) and lo(one)[-1] < lo(one)[0]


def test_two_rank_multivae_drop_in_trains_as_replicas(tmp_path):
    """conf/MultiVAE.properties on two ranks: data-parallel replicas (each rank its share of every global batch, one
    all-reduce of the gradients per step).  The sampling noise of a replica run is a different, equally distributed draw
    than one process's, so the check is the run itself: one log, the loss falls, the evaluation (sharded over users)
    prints a line of the reference's format with a model that has learnt."""
    args = ["--recommender=MultiVAE", "--epochs=12", "--batch_size=64", "--learning_rate=0.01", "--verbose=6"]
    one = _run(str(tmp_path / "one"), 1, [], args, "MultiVAE", raw=True)
    two = _run(str(tmp_path / "two"), 2, [], args, "MultiVAE", raw=True)
    for text in (one, two):
        lo = [float(x) for x in re.findall(r"\[iter \d+ : loss : ([0-9.]+),", text)]
        ev = re.findall(r"epoch (\d+):\t(.+)", text)
        assert len(lo) == 12 and lo[-1] < lo[0] and [e[0] for e in ev] == ["6", "12"]
        vals = [float(x) for x in ev[-1][1].split("\t")]
        assert len(vals) == 10 and max(vals) > 0.05
    l1 = [float(x) for x in re.findall(r"\[iter \d+ : loss : ([0-9.]+),", one)]
    l2 = [float(x) for x in re.findall(r"\[iter \d+ : loss : ([0-9.]+),", two)]
    assert abs(l1[-1] - l2[-1]) <= 0.1 * abs(l1[-1])               # the same model, another draw of the noise
