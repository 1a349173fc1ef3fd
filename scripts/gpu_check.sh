#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, smoke, bench, rocprof summary.
# Everything is written under gpurun_out/ so it is merged back.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/$(date +%H%M%S)
mkdir -p "$OUT"
echo "== device" | tee "$OUT/summary.txt"
rocm-smi --showproductname 2>/dev/null | head -8 | tee -a "$OUT/summary.txt"
PHASES="${1:-tests smoke bench prof}"
for ph in $PHASES; do
  case $ph in
    tests)
      echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
      timeout 600 python -m pytest tests -m gpu -q -x --maxfail=50 -p no:cacheprovider --tb=short 2>&1 | tail -150 > "$OUT/pytest_gpu.log"
      tail -40 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt" ;;
    testsall)
      echo "== pytest -m gpu (no -x)" | tee -a "$OUT/summary.txt"
      timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -400 > "$OUT/pytest_gpu.log"
      tail -60 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt" ;;
    smoke)
      echo "== smoke" | tee -a "$OUT/summary.txt"
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -30 | tee -a "$OUT/summary.txt" ;;
    bench)
      echo "== bench" | tee -a "$OUT/summary.txt"
      timeout 420 python bench.py --gpus 1 --steps 300 --warmup 30 ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
      tail -5 "$OUT/bench.err" | tee -a "$OUT/summary.txt"; cat "$OUT/bench.json" | tee -a "$OUT/summary.txt" ;;
    prof)
      echo "== rocprofv3 kernel stats" | tee -a "$OUT/summary.txt"
      ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err" )
      find "$OUT/prof" -name "*kernel_stats*" | head -3 | tee -a "$OUT/summary.txt"
      f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && head -25 "$f" | tee -a "$OUT/summary.txt"
      # keep the merge-back small: drop the raw per-dispatch trace, keep stats
      find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete ;;
  esac
done
echo "== done" | tee -a "$OUT/summary.txt"
