#!/bin/bash
# Average shader clock while a kernel runs: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) of
# every kernel of one bench run next to its duration from the kernel trace of the same pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/clock
mkdir -p "$OUT"
( cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OLDPWD/$OUT/run" -o b -- python "$OLDPWD/bench.py" --gpus 1 --steps 20 --warmup 2 --no-cpu-baseline --no-mf > /dev/null 2> "$OLDPWD/$OUT/err.txt" )
c=$(find "$OUT/run" -name "*counter_collection.csv" | head -1)
[ -z "$c" ] && { tail -5 "$OUT/err.txt"; exit 1; }
python - "$c" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        dur = float(row.get("End_Timestamp", 0) or 0) - float(row.get("Start_Timestamp", 0) or 0)
        a = agg[k]; a[0] += 1; a[1] += float(row["Counter_Value"]); a[2] += dur
print("%-60s %6s %12s %10s %9s" % ("kernel", "calls", "cycles/XCD", "us", "GHz"))
for k, (n, cyc, dur) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:12]:
    if dur > 0:
        print("%-60s %6d %12.0f %10.1f %9.2f" % (k, n, cyc / n / 8, dur / n / 1e3, (cyc / 8) / dur))
PY
rm -rf "$OUT/run"
