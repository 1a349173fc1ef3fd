#!/bin/bash
# SQ counters of one kernel (substring match) under any python command, separate passes per group.
# usage (through gpurun): bash scripts/pmc_kernel.sh <kernel substring> <out tag> -- python scripts/xyz.py ...
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
KSUB=$1; TAG=$2; shift 3
OUT=$R/gpurun_out/r04/pmc_$TAG
mkdir -p "$OUT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/g$i" -o b -- "$@" > /dev/null 2> "$OUT/g$i.err" )
  f=$(find "$OUT/g$i" -name "*counter_collection.csv" | head -1)
  echo "== $group"
  if [ -n "$f" ]; then python - "$f" "$KSUB" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if sys.argv[2] not in row.get("Kernel_Name", ""):
            continue
        k = row.get("Counter_Name", "")
        agg[k][0] += 1
        agg[k][1] += float(row.get("Counter_Value", 0) or 0)
for k, (n, v) in sorted(agg.items()):
    print("   %-40s launches=%3d  avg/launch=%.5g" % (k, n, v / n))
PY
  else tail -3 "$OUT/g$i.err"; fi
  rm -rf "$OUT/g$i"
done < <(if [ -n "${PMC_GROUPS_FILE:-}" ]; then cat "$PMC_GROUPS_FILE"; else cat <<GROUPS_EOF
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS
SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES
GROUPS_EOF
fi)
