#!/bin/bash
# rocprof kernel durations of the BPR-MF step variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/mf_fused; mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/p" -o b -- python "$OLDPWD/scripts/exp_mf_fused.py" > "$OLDPWD/$OUT/run.txt" 2>/dev/null )
cat $OUT/run.txt | tail -12
python - "$OUT/p" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+)(<[^>]*>)?\(", n)
    print("%-50s calls %6s avg %8.2f us" % (m.group(0)[:50] if m else n[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
