#!/bin/bash
# r05: the lane-group schedule's per-sub-list cost (rows are dealt to workgroups by nnz + ENTCOST * sub-lists): step time
# of the bench workload per value
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in 0 2 4 6 8 12 16; do
  NEUREC_SPMM_ENTCOST=$c timeout 200 python bench.py --no-config4 --no-config5 --no-cpu-baseline --no-mf --no-eval --steps 400 --warmup 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ENTCOST=$c ms/step %.4f  spmm us %.2f' % (d['ms_per_step'], d['roofline']['us_per_launch']))"
done
