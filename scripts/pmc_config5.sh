#!/bin/bash
# SQ counters of the config-5 kernels (separate --pmc passes, kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc5; mkdir -p $OUT
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/$OUT/$tag" -o b -- python "$OLDPWD/scripts/bench_config5.py" > /dev/null 2> "$OLDPWD/$OUT/$tag.err" )
done
python - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if not any(t in k for t in ("vae_dwp1", "vae_dg1_mfma", "vae_softmax_stats", "spmm_blocked_kernel<false, 16, 8, 16", "ngcf_layer")):
            continue
        a = agg[k][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for k, cs in agg.items():
    print(k)
    print("   " + "  ".join("%s=%.3g" % (c, v[1] / max(v[0], 1)) for c, v in sorted(cs.items())))
PY
