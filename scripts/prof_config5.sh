#!/bin/bash
# rocprof kernel table of the NGCF / Mult-VAE steps (BASELINE configs[4])
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/config5; mkdir -p $OUT
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/p" -o b -- python "$OLDPWD/scripts/bench_config5.py" > "$OLDPWD/$OUT/run.txt" 2>/dev/null )
tail -5 $OUT/run.txt
python - "$OUT/p" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+)(<[^>]*>)?\(", n)
    print("%-60s calls %6s avg %8.2f us  %5s%%" % (m.group(0)[:60] if m else n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
