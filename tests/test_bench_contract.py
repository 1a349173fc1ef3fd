"""bench.py's output contract, checked on the committed bench line of the round
(profiles/r06_bench.json) and on the argument parser — no GPU needed."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r06_bench.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_committed_bench_line_carries_every_contract_field():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "triplets/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value and ms_per_step describe the same run: steps * global batch / time
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is not None and r["traffic"] > r["bytes_per_launch"]   # misses re-read rows
    # the PMC figure belongs to the SpMM sources this tree holds (bench.py nulls a stale one)
    import hashlib
    h = hashlib.sha256()
    for name in ("spmm_blocked.hip", "spmm.hip"):
        with open(os.path.join(ROOT, "neurec_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    with open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")) as f:
        assert json.load(f)["_spmm_sources_sha16"] == h.hexdigest()[:16]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1
    # SURVEY 8d's three CPU legs: the step (port), the reference's own sampler and evaluator
    assert c["sampler"]["kind"] == "reference" and c["sampler"]["unit"] == "triplets/s"
    assert c["eval"]["kind"] == "reference" and c["eval"]["unit"] == "users/s"


def test_committed_bench_line_says_what_was_inside_the_timed_region():
    d = _line()
    t = d["timed_region"]
    assert t["steps"] == d["steps"] and t["sampler_launches"] >= 0 and t["batch_plan_launches"] >= 0
    assert "data_note" in d and "synthetic" in d["data_note"]
    e = d["eval"]
    assert e["ndcg10_oracle_absdiff"] == 0.0                       # metric value pinned on the reference's evaluator
    r = e["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert 0 < e["roofline_topk"]["frac"] < 1
    # the search arithmetic is named, and it certified every row (a redone row would be a full fp32 row in the timing)
    assert e["search"] in ("int8", "bf16", "fp32") and r.get("arith", e["search"]) == e["search"]
    assert e["rows_redone_for_ties"] == 0 and r["unit"] == ("TOP/s" if e["search"] == "int8" else "TFLOP/s")
    with open(os.path.join(ROOT, "profiles", "r06_bench_driver_line.json")) as f:
        compact = json.loads(f.read().strip().splitlines()[-1])["roofline"]
    for k in ("eval_search", "eval_search_unit", "eval_fp32_roof_ratio", "eval_fp32_loop_ms", "eval_rows_redone", "eval_rank_ms"):
        assert k in compact, k
    assert compact["eval_search"] == e["search"] and compact["eval_ndcg10_oracle_absdiff"] == 0.0
    m = d["mf"]
    assert "lazy" in m["optimizer"] and m["ms_per_step"] < m["sweep_ms_per_step"]


def test_bench_metric_is_the_north_star_metric():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    d = _line()
    text = json.dumps(base).lower()
    assert "triplets" in text and "triplets" in d["metric"].lower()
    assert "lightgcn" in d["metric"].lower() and "gowalla" in d["config"]["workload"].lower()


def test_bench_cli_has_the_contract_flags_and_safe_defaults():
    with open(os.path.join(ROOT, "bench.py")) as f:
        src = f.read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert re.search(r'add_argument\(\s*"%s"' % flag, src), flag
    m = re.search(r'add_argument\(\s*"--gpus"[^)]*default=(\d+)', src)
    assert m and int(m.group(1)) == 1
    # nothing measured may touch the reference tree or the oracle outside the cpu_baseline legs (bench_legs.py holds them)
    with open(os.path.join(ROOT, "bench_legs.py")) as f:
        legs = f.read()
    assert "/root/reference" not in src and "/root/reference" not in legs
    # `oracle` is imported inside the CPU-baseline code only: never at module level of either file
    for text in (src, legs):
        assert not any(l.startswith(("import oracle", "from oracle")) for l in text.splitlines())
