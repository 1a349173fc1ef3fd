cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
cat > /tmp/mf1.py <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neurec_amd import engine as E, synth
from neurec_amd.trainer import BprEpochSampler, MFEngine
tr, _ = synth.interactions("gowalla", seed=2018)
U, I = tr.shape
trc = E.DeviceCSR.from_scipy(tr)
smp = BprEpochSampler(trc, I, batch_size=512, seed=2018, plan_users=U)
batches = [b for b in smp.batches() if b[0].numel() == 512][:400]
loss = torch.zeros(2, device="cuda")
rs = np.random.RandomState(2017)
mf = MFEngine((rs.randn(U, 64) * 0.01).astype(np.float32), (rs.randn(I, 64) * 0.01).astype(np.float32), 0.001, 0.0, 512, lazy=True, lazy_period=int(sys.argv[1]))
for b in batches: mf.step(b[0], b[1], b[2], loss, plan=b.plan, next_plan=b.next_plan)
torch.cuda.synchronize()
PY
for per in 16; do
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm$per -o m -- python /tmp/mf1.py $per > /dev/null 2>&1 )
echo "== period $per"; f=$(find /tmp/pm$per -name "*kernel_stats.csv" | head -1); python -c "import csv,sys; [print(r[\"Name\"][:60], r[\"Calls\"], r[\"AverageNs\"], r[\"MinNs\"], r[\"MaxNs\"]) for r in list(csv.DictReader(open(sys.argv[1])))[:4]]" "$f"
done
