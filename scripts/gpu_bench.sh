#!/bin/bash
# usage (through gpurun): bash scripts/gpu_bench.sh <tag> [bench.py args...]  -> gpurun_out/r03/<tag>.json
tag=$1; shift
mkdir -p gpurun_out/r03
python bench.py "$@" > gpurun_out/r03/$tag.json 2> gpurun_out/r03/$tag.err
echo "rc=$?"; tail -c 1500 gpurun_out/r03/$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/r03/$tag.json"))
print("value %.0f ms/step %.4f spmm us %.2f frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["us_per_launch"], d["roofline"]["frac"]))
print("epoch_amortised", d.get("epoch_amortised", {}).get("value"))
if d.get("eval"): print("eval ms", d["eval"]["ms"], "absdiff", d["eval"].get("ndcg10_oracle_absdiff"), "score frac", d["eval"].get("roofline", {}).get("frac"))
if d.get("mf"): print("mf ms", d["mf"]["ms_per_step"], "absdiff", d["mf"].get("ndcg10_oracle_absdiff"))
for k in ("ngcf","multivae","config4"):
    v = d.get(k)
    if v: print(k, "ms/step", v["ms_per_step"], "roofline frac", v["roofline"]["frac"], "us", v["roofline"]["us_per_launch"], "cpu", (v.get("cpu_baseline") or {}).get("value"), v.get("setup_seconds"))
PY
python - <<PY
import json
d=json.load(open("gpurun_out/r03/$tag.json"))
w=(d.get("multivae") or {}).get("wide")
if w: print("multivae wide", w["p_dim"], "ms/step", w["ms_per_step"], "item layer us", w["roofline"]["us_per_step"], "TFLOP/s", w["roofline"]["achieved"], "frac", w["roofline"]["frac"])
PY
python - <<PY
import json
d=json.load(open("gpurun_out/r03/$tag.json"))
w=(d.get("ngcf") or {}).get("wide")
if w: print("ngcf wide", w["dim"], w["layers"], "ms/step", w["ms_per_step"], "spmm us", w["roofline"]["us_per_launch"], "frac", w["roofline"]["frac"])
PY
