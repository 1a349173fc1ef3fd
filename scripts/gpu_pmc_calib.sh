#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_calib; mkdir -p "$OUT"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OLDPWD/$OUT" -o c -- python "$OLDPWD/scripts/exp_pmc_calib.py" > "$OLDPWD/$OUT/stdout.txt" 2> "$OLDPWD/$OUT/err.txt" )
cat "$OUT/stdout.txt" | grep -v amdgpu
python - "$(find $OUT -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "spmm_item" in r["Kernel_Name"]]
print("FETCH_SIZE per spmm launch (KB):", [round(v) for v in vals])
PY
