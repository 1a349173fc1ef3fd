#!/bin/bash
# Cache / TLB / stall counters of the lane-group SpMM kernel: one --pmc pass per counter group
# (kernel-trace only), averaged per launch.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_cache
mkdir -p "$OUT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  ( cd /tmp && timeout 150 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OLDPWD/$OUT/g$i" -o b -- python "$OLDPWD/scripts/exp_spmm_only.py" > /dev/null 2> "$OLDPWD/$OUT/g$i.err" )
  f=$(find "$OUT/g$i" -name "*counter_collection.csv" | head -1)
  echo "== $group"
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if "spmm_blocked" not in row.get("Kernel_Name", ""):
            continue
        k = row.get("Counter_Name", "")
        agg[k][0] += 1
        agg[k][1] += float(row.get("Counter_Value", 0) or 0)
for k, (n, v) in sorted(agg.items()):
    print("   %-40s launches=%3d  avg/launch=%.4g" % (k, n, v / n))
PY
  else tail -3 "$OUT/g$i.err"; fi
  rm -rf "$OUT/g$i"
done <<'GROUPS'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCC_TAG_STALL_sum TCC_BUSY_avr
GRBM_GUI_ACTIVE FETCH_SIZE
GROUPS
