#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for k in ${@:-1 2 4 5 6}; do
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python $R/scripts/exp_vae_fused.py $k 2>&1 | grep -v "^W2\|^E2" | tail -1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    if "vae_" in r["Name"]:
        print("   %9.1f us avg  %s" % (float(r["AverageNs"]) / 1e3, r["Name"][:80]))
PY
done
