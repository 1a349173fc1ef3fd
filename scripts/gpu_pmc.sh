#!/bin/bash
# HBM-side traffic of the training step's kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (kernel-trace only), as /opt/skills/guides/MI355X_MICROARCH.md prescribes, over the bench command.
# Writes gpurun_out/pmc/pmc_traffic.json (copy to profiles/rNN_pmc_traffic.json): per kernel the
# per-launch averages, FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), stamped with the
# hash of the SpMM sources so that bench.py can tell when the figure is stale.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc
mkdir -p "$OUT"
export C4SCALE=${C4SCALE:-0.25}
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/$c" -o b -- python "$OLDPWD/bench.py" --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-eval --no-mf --no-config5 --config4-scale ${C4SCALE:-0.25} > /dev/null 2> "$OLDPWD/$OUT/$c.err" )
done
python - "$OUT" <<'PY'
import collections, csv, glob, hashlib, json, os, sys
out = sys.argv[1]
def short(k):
    if k.startswith("void "): k = k[5:]
    if k.startswith("(anonymous namespace)::"): k = k[len("(anonymous namespace)::"):]
    return k.split("(")[0]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                agg[k][0] += 1
                agg[k][1] += float(row.get("Counter_Value", 0) or 0)
    for k, (n, v) in agg.items():
        res[k][c.lower() + "_kb"] = v / n
        res[k]["launches"] = n
h = hashlib.sha256()
for name in ("spmm_blocked.hip", "spmm.hip"):
    with open(os.path.join("neurec_amd", "csrc", name), "rb") as f:
        h.update(f.read())
doc = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel-trace only) over "
               "`python bench.py --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-eval --no-mf --no-config5 --config4-scale "
               + os.environ.get("C4SCALE", "0.25") + "` "
               "(scripts/gpu_pmc.sh); per-launch averages; FETCH_SIZE doubled (gfx950 tallies 128-B requests at "
               "64 B: guide + profiles/r01_pmc_calibration_gather.txt), WRITE_SIZE as reported. Counts L2->fabric "
               "requests, Infinity-Cache hits included: an upper bound on HBM bytes.",
       "_spmm_sources_sha16": h.hexdigest()[:16], "kernels": {}}
for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("fetch_size_kb", 0))):
    if "fetch_size_kb" in v and "write_size_kb" in v:
        v["traffic_bytes_per_launch"] = int((2 * v["fetch_size_kb"] + v["write_size_kb"]) * 1000)
        doc["kernels"][k] = v
with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
    json.dump(doc, f, indent=1)
for k, v in list(doc["kernels"].items())[:10]:
    print("%-64s launches=%4d fetch(x2)=%.1f MB write=%.1f MB" % (k[:64], v["launches"], 2 * v["fetch_size_kb"] / 1e3, v["write_size_kb"] / 1e3))
PY
find "$OUT" -name "*.csv" -size +5M -delete
