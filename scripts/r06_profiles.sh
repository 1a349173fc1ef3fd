#!/bin/bash
# The round's records on the GPU box: (1) the plain bench line (compact + full), (2) rocprofv3 --kernel-trace --stats of
# the same command, (3) the PMC traffic passes (scripts/gpu_pmc.sh), (4) the config-5 kernel table.  -> gpurun_out/r06/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
echo "== PMC traffic (first: the bench line reports it when it matches the sources)"
C4SCALE=${C4SCALE:-1.0} bash scripts/gpu_pmc.sh 2>&1 | tail -12
cp gpurun_out/pmc/pmc_traffic.json $OUT/pmc_traffic.json
cp gpurun_out/pmc/pmc_traffic.json profiles/r06_pmc_traffic.json   # (so that the bench run below reports the traffic of THESE sources)
echo "== bench (plain)"
NEUREC_BENCH_FULL=$OUT/final_bench_full.json timeout 900 python bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err
tail -2 $OUT/final_bench.err; head -c 600 $OUT/final_bench.json; echo
echo "== rocprofv3 kernel stats of the bench command"
( cd /tmp && NEUREC_BENCH_FULL=/tmp/prof_full.json timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err" )
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv && head -12 "$f" | cut -c1-160
find "$OUT/prof" -name "*kernel_trace.csv" -delete
find "$OUT/prof" -name "*.csv" -size +5M -delete
echo "== config 5 kernel table"
bash scripts/prof_config5.sh 2>&1 | tail -45 > $OUT/config5_kernels.txt
f=$(find gpurun_out/config5/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/config5_kernel_stats.csv
tail -5 $OUT/config5_kernels.txt
echo "== evaluation-only kernel table (no torch / rocPRIM kernel may appear between the evaluator's first and last launch)"
bash scripts/prof_py.sh eval scripts/exp_eval_time.py 2>&1 | tail -30 > $OUT/eval_kernels.txt
tail -25 $OUT/eval_kernels.txt
