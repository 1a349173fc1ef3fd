#!/bin/bash
# evaluation time of the pruned design vs the user batch size (one batch vs several, overlapped on two streams)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for eb in 32768 16384 8192 4096; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mf --eval-batch $eb 2>/dev/null | EB=$eb python -c '
import json, os, sys
d = json.loads(sys.stdin.read()); e = d["eval"]
print("eval-batch", os.environ["EB"], "ms", round(e["ms"], 3), "Musers/s", round(e["users_per_sec"] / 1e6, 2), "scoring ms", round(e["roofline"]["ms"], 3), "level2 ms", round(e["roofline_topk"]["ms"], 3))'
done
