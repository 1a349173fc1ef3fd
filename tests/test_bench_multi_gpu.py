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
           "--no-cpu-baseline", "--no-mf", "--config4-scale", "0.002", "--full-line"] + (["--dp-mode", mode] if mode else [])
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("mode,port", [("replicated", 29611), ("triplets", 29612), ("allreduce", 29613),
                                       ("rowshard", 29614), (None, 29615), ("colshard", 29616)])
def test_two_rank_bench_line(mode, port):
    d = _run(mode, port)
    # what of the step every rank repeats is said in the line; the default N>1 mode partitions ALL of it (column-sharded
    # tables, one all-gather of partial inner products per step); the fully redundant modes are opt-in
    assert d["dist_backend"] == "gloo" and d["rccl_ranks"] == 0          # this test box has one GPU: no RCCL here
    if mode in (None, "colshard"):
        assert d["config"]["parallelism"].startswith("colshard2 (every rank holds 32 of the 64")
        assert d["redundant_compute"] is False and "computed twice" in d["redundant_compute_note"]
        assert d["roofline"]["kernel"].endswith("32, false>")             # the rank's kernels run at its 32 columns
    if mode == "allreduce":
        assert "one all-reduce" in d["config"]["parallelism"] and d["redundant_compute"].startswith("the propagation")
    if mode in ("replicated", "triplets"):
        assert d["redundant_compute"].startswith("everything")
    if mode == "rowshard":
        assert d["redundant_compute"] is False
    else:
        assert d["same_global_batch_on_1gpu"]["global_batch"] == 512 and d["same_global_batch_on_1gpu"]["value"] > 0
    leg = d["rowshard_config4_law"]                                        # every N>1 line carries the row-shard leg
    assert leg["ranks"] == 2 and leg["ms_per_step"] > 0 and leg["hop"] == "sliced" and leg["exchange"]["hop_ms"] > 0
    assert leg["eval"]["users_per_sec"] > 0 and leg["eval"]["n_users"] > 0   # ... with its sharded evaluation (r06)
    if mode in (None, "colshard"):                                         # strong scaling next to the weak figure
        assert d["strong_scaling"]["global_batch"] == 256 and d["strong_scaling"]["value"] > 0
        assert "rowshard" in d["north_star_partition"]
    red = d["rowshard_config4_law_reduce"]                                 # ... and its reduced-exchange form
    assert red["hop"] == "reduce" and red["ms_per_step"] > 0
    assert red["exchange"]["received_per_rank_bytes_per_hop"] < leg["exchange"]["received_per_rank_bytes_per_hop"]
    assert abs(leg["scale"] - 0.002 * 2 / 8) < 1e-12
    mode = mode or "colshard"
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 512
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert str(2) in d["config"]["parallelism"]
    assert all(x == x and abs(x) < 1e9 for x in d["final_loss"])           # finite
    if mode != "rowshard":
        assert d["eval"]["n_users"] > 0 and 0.0 <= d["eval"]["ndcg@10"] <= 1.0
    assert d["roofline"]["frac"] > 0


@pytest.mark.parametrize("extra", [["--scale", "0.05", "--batch", "256"],
                                   ["--shape", "config4", "--scale", "0.0005", "--dp-mode", "rowshard", "--dim", "128",
                                    "--batch", "256"]], ids=["gowalla-small", "config4-small"])
def test_one_rank_bench_line(extra):
    """The default single-process line (training + evaluation + BPR-MF legs) and the config-4 path
    (device-generated graph, row-sharded engine) on small shapes: contract fields present and finite."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--full-line"] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None
    assert all(x == x and abs(x) < 1e9 for x in d["final_loss"])
    if "config4" not in extra:
        assert "ndcg10_oracle_absdiff" not in d["eval"]              # that comparison belongs to the CPU-baseline leg
        assert d["mf"]["ms_per_step"] > 0 and d["eval"]["roofline"]["frac"] > 0
        assert d["mf"]["epoch"]["triplets"] == d["mf"]["interactions"]          # a whole epoch, short last batch inside
        # the headline's whole timed epoch: every triplet of the stream, the sampler's launch inside
        assert d["epoch_timed"]["triplets"] == d["mf"]["interactions"] and d["epoch_timed"]["sampler_launches"] == 1
        assert d["epoch_timed"]["short_last_batch"] == (d["mf"]["interactions"] % 256 != 0) and d["epoch_timed"]["value"] > 0
    else:
        assert d["eval"]["users_per_sec"] > 0 and d["eval"]["n_users"] > 0       # config 4 has an evaluation leg (r06)


def test_default_line_is_the_compact_one():
    """Without --full-line the printed line is the compact form the driver's parser keeps whole: the
    contract keys, `roofline` with the secondary legs as flat numbers, no prose; the full line goes to a file."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--scale", "0.05", "--batch", "256", "--no-config4", "--no-config5"]
    full = os.path.join(ROOT, "gpurun_out", "test_full_line.json")
    os.makedirs(os.path.dirname(full), exist_ok=True)
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=280, env=dict(os.environ, NEUREC_BENCH_FULL=full))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4000
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    r = d["roofline"]
    assert r["eval_users_per_sec"] > 0 and r["mf_triplets_per_sec"] > 0 and 0 < r["frac"] < 1
    assert all(not isinstance(v, (dict, list)) for v in r.values())
    with open(full) as f:
        assert json.load(f)["eval"]["roofline"]["kernel"]              # the full objects live in the file


def test_plain_shell_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (WORLD_SIZE unset): bench.spawn_ranks starts the ranks itself —
    one process per rank, a FILE-STORE rendezvous (no port is picked and lost before the ranks bind it) — and
    rank 0's stdout is the ONE line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NEUREC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--scale", "0.05",
           "--batch", "256", "--no-cpu-baseline", "--no-mf", "--no-config4", "--no-config5", "--full-line"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist_backend"] == "gloo" and d["config"]["global_batch"] == 512
    assert d["strong_scaling"]["global_batch"] == 256 and d["value"] > 0
