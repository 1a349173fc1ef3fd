#!/bin/bash
# rocprofv3 kernel table of the wide Mult-VAE step -> gpurun_out/r03/wide_vae_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wv -o wv -- python $R/scripts/prof_wide_vae.py 20 2>&1 | tail -8
f=$(find /tmp/wv -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r03/wide_vae_kernel_stats.csv
head -30 "$f"
