#!/bin/bash
# kernel breakdown of the LightGCN step at the global batch the 8-GPU id-exchange mode steps on (8 x 1,024)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/bigbatch; mkdir -p $OUT
for B in ${1:-8192}; do
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/p$B" -o b -- python "$OLDPWD/bench.py" --batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-mf --no-eval > "$OLDPWD/$OUT/b$B.json" 2>/dev/null )
python - "$OUT/p$B" $B <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
print("B =", sys.argv[2])
for r in list(csv.DictReader(open(f)))[:9]:
    print("%-60s calls %5s avg %8.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
python -c "import json; d=json.load(open('$OUT/b$B.json')); print('ms/step', d['ms_per_step'], 'triplets/s', d['value'])"
done
