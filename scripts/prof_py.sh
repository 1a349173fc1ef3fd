#!/bin/bash
# rocprofv3 kernel table of any script: bash scripts/prof_py.sh <tag> <script.py> [args]  -> gpurun_out/${ROUND:-r05}/<tag>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
mkdir -p $R/gpurun_out/${ROUND:-r05}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python $R/$@ 2>&1 | grep -v "^W2\|^E2" | tail -5
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${ROUND:-r05}/${tag}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%9.1f us avg  calls %5d  total %8.2f ms  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, r["Name"][:90]))
PY
