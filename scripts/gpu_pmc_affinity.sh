#!/bin/bash
# L2 counters of the full-pass SpMM kernels under the affinity-schedule knobs: one --pmc pass per
# (config, counter group), kernel-trace only, averaged per launch.  Usage: gpu_pmc_affinity.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_affinity
mkdir -p "$OUT"
CONFIGS=${PMC_CONFIGS:-0:0 1:1000000000 1:5500000 1:4000000 1:2621440}
for cfg in $CONFIGS; do
  aff=${cfg%%:*}; blk=${cfg##*:}
  echo "#### NEUREC_SPMM_AFFINITY=$aff NEUREC_SPMM_AFF_BLOCK=$blk" | tee -a "$OUT/summary.txt"
  for group in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    ( cd /tmp && NEUREC_SPMM_AFFINITY=$aff NEUREC_SPMM_AFF_BLOCK=$blk timeout 150 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OLDPWD/$OUT/g" -o b -- python "$OLDPWD/scripts/exp_affinity.py" > "$OLDPWD/$OUT/run.log" 2> "$OLDPWD/$OUT/g.err" )
    f=$(find "$OUT/g" -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python - "$f" <<'PY' | tee -a "$OUT/summary.txt"
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row.get("Kernel_Name", "")
        if "spmm_blocked_kernel" not in name and "spmm_affinity_kernel" not in name:
            continue
        k = (name.split("<")[0].split("::")[-1], row.get("Counter_Name", ""))
        agg[k][0] += 1
        agg[k][1] += float(row.get("Counter_Value", 0) or 0)
for k, (n, v) in sorted(agg.items()):
    print("   %-24s %-22s launches=%3d  avg/launch=%.4g" % (k[0], k[1], n, v / n))
PY
    else tail -3 "$OUT/g.err"; fi
    tail -1 "$OUT/run.log" | tee -a "$OUT/summary.txt"
    rm -rf "$OUT/g"
  done
done
