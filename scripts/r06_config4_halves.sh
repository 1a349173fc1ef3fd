#!/bin/bash
# The two halves of the config-4 pass: times (plain run) and L2 -> fabric bytes per launch (FETCH_SIZE pass).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_halves; mkdir -p $OUT
S=${1:-1.0}
timeout 600 python scripts/exp_config4_halves.py $S > $OUT/halves.json 2> $OUT/halves.err; cat $OUT/halves.json
( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OLDPWD/$OUT/pmc" -o h -- python "$OLDPWD/scripts/exp_config4_halves.py" $S > /dev/null 2> "$OLDPWD/$OUT/pmc.err" )
python - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if "spmm" in k:
                agg[k.split("(")[0][-60:]].append(float(row.get("Counter_Value", 0) or 0))
with open(os.path.join(out, "fetch.txt"), "w") as fo:
    for k, v in agg.items():
        # launches in program order: Mu x4 (1 warm + 3), Mp x4, full x4 — all the same template: list every launch
        line = "%s: %d launches, FETCH_SIZE x2 per launch (GB): %s" % (k, len(v), " ".join("%.1f" % (2 * x / 1e6) for x in v))
        print(line); fo.write(line + "\n")
PY
find "$OUT" -name "*.csv" -size +5M -delete
