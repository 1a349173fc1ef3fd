#!/bin/bash
# HBM traffic counters for the SpMM kernel: separate --pmc passes (kernel-trace only), per the guide.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc
mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/$c" -o b -- python "$OLDPWD/bench.py" --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-eval --no-mf > /dev/null 2> "$OLDPWD/$OUT/$c.err" )
  f=$(find "$OUT/$c" -name "*counter_collection.csv" | head -1)
  echo "== $c : $f"
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name", "")
        if k.startswith("void "):
            k = k[5:]
        if k.startswith("(anonymous namespace)::"):
            k = k[len("(anonymous namespace)::"):]
        k = k.split("(")[0][:70]
        agg[k][0] += 1
        agg[k][1] += float(row.get("Counter_Value", 0) or 0)
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%-70s calls=%5d  avg=%.1f (counter units, KB per FETCH/WRITE_SIZE)" % (k, n, v / n))
PY
  find "$OUT/$c" -name "*.csv" -size +5M -delete
done
