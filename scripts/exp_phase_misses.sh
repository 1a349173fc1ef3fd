cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for blk in 0 5500000 4000000 2621440; do
  echo "#### base kernel, NEUREC_SPMM_BLOCK_BYTES=$blk"
  ( cd /tmp && NEUREC_SPMM_AFFINITY=0 NEUREC_SPMM_BLOCK_BYTES=$blk timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmcx -o b -- python "$OLDPWD/scripts/exp_affinity.py" > /tmp/run.log 2> /tmp/g.err )
  f=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row.get("Kernel_Name", "")
        if "spmm_blocked_kernel" not in name: continue
        k = row.get("Counter_Name", "")
        agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0) or 0)
for k, (n, v) in sorted(agg.items()): print("   %-22s launches=%3d  avg/launch=%.4g" % (k, n, v / n))
PY
  rm -rf /tmp/pmcx
  NEUREC_SPMM_AFFINITY=0 NEUREC_SPMM_BLOCK_BYTES=$blk python scripts/exp_affinity.py 2>&1 | tail -1
done
