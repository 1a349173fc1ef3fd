"""bench.py's N>1 launch line, as the driver issues it, on the one GPU of the test box: two ranks
share the device over gloo (NEUREC_DIST_BACKEND), every --dp-mode, a small graph.  Checks the
contract fields of the printed line and that every mode trains the same model: the final losses
of the modes that step on the same global batch are equal."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(mode, port):
    env = dict(os.environ, NEUREC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--scale", "0.05", "--batch", "256",
           "--no-cpu-baseline", "--no-mf", "--dp-mode", mode]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("mode,port", [("replicated", 29611), ("triplets", 29612), ("allreduce", 29613),
                                       ("rowshard", 29614)])
def test_two_rank_bench_line(mode, port):
    d = _run(mode, port)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 512
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert str(2) in d["config"]["parallelism"]
    assert all(x == x and abs(x) < 1e9 for x in d["final_loss"])           # finite
    if mode != "rowshard":
        assert d["eval"]["n_users"] > 0 and 0.0 <= d["eval"]["ndcg@10"] <= 1.0
    assert d["roofline"]["frac"] > 0
