#!/bin/bash
# kernel table of the narrow Mult-VAE step, fused decoder and (NEUREC_VAE_DECODER=slab) the slab form -> gpurun_out/r04/
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for mode in fused slab; do
rm -rf /tmp/pp
NEUREC_VAE_DECODER=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python $R/scripts/prof_narrow_vae.py 2>&1 | grep -v "^W2\|^E2" | tail -2
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r04/narrow_vae_${mode}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%9.1f us avg  calls %5d  total %8.2f ms  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, r["Name"][:110]))
PY
done
