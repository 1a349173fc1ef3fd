"""BASELINE configs[4]: NGCF and Mult-VAE on the gowalla-shaped synthetic interactions, 1 GPU.
Prints one JSON object with per-step times (HIP-event timed) and throughput for both models."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurec_amd import engine as E, synth
from neurec_amd.graph import ngcf_adjacency, transpose_csr
from neurec_amd.trainer import BprEpochSampler, FullRankEvaluator, MultiVAEEngine, NGCFEngine
from neurec_amd.util.tool import get_initializer

tr, te = synth.interactions("gowalla", seed=2018)
U, I = tr.shape
dev = torch.device("cuda")
trc, tec = E.DeviceCSR.from_scipy(tr), E.DeviceCSR.from_scipy(te)
out = {"shape": {"users": U, "items": I, "interactions": int(tr.nnz)}}


def timed(fn, n, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

# ---- NGCF: conf/NGCF.properties (embedding 16, layers [16,16], B=512, norm adjacency, dropout 0.1)
A = ngcf_adjacency(tr, "norm")
w = get_initializer("xavier_normal", 0.01, seed=2018)
e = get_initializer("xavier_normal", 0.01, seed=2017)
table = np.concatenate([e([U, 16]), e([I, 16])])
weights = [(w([16, 16]), w([1, 16]), w([16, 16]), w([1, 16])) for _ in range(2)]
ng = NGCFEngine(A, transpose_csr(A), U, I, table, weights, 0.001, 0.0, 0.1, 512)
sampler = BprEpochSampler(trc, I, neg_num=1, batch_size=512, shuffle=True, seed=2018, plan_users=U)
batches = [b for b in sampler.batches() if b[0].numel() == 512][:200]
loss = torch.zeros(2, device=dev)
it = iter(batches * 10)
def ngcf_step():
    b = next(it)
    ng.step(b[0], b[1], b[2], loss, plan=b.plan)
ms = timed(ngcf_step, 150, 20)
ev = FullRankEvaluator(trc, tec, [1, 2, 4, 3, 5], 20, batch_rows=16384)
users = torch.from_numpy(np.flatnonzero(np.diff(te.indptr) > 0).astype(np.int32)).to(dev)
def ngcf_eval():
    eu, ei = ng.final_embeddings()
    return ev.evaluate_factors(eu.contiguous(), ei.contiguous(), users)
ngcf_eval(); torch.cuda.synchronize(); t0 = time.perf_counter(); m = ngcf_eval(); torch.cuda.synchronize()
out["ngcf"] = {"ms_per_step": ms, "triplets_per_sec": 512 / ms * 1e3, "batch": 512, "dim": 16, "layers": [16, 16],
               "eval_users_per_sec": users.numel() / (time.perf_counter() - t0), "ndcg@10": float(m[2 * 20 + 9]),
               "adjacency_nnz": int(A.nnz)}

# ---- Mult-VAE: conf/MultiVAE.properties (p_dim [16,32], B=512, tanh, keep 0.8)
wi = get_initializer("xavier_normal", 0.01, seed=2017); bi = get_initializer("tnormal", 0.01, seed=2018)
z, h = 16, 32
params = {"Wq0": wi([I, h]), "bq0": bi([h]), "Wq1": wi([h, 2 * z]), "bq1": bi([2 * z]), "Wp0": wi([z, h]),
          "bp0": bi([h]), "Wp1t": np.ascontiguousarray(wi([h, I]).T), "bp1": bi([I])}
vae = MultiVAEEngine(trc, I, params, 0.001, 0.0, "tanh", 512)
perm = torch.from_numpy(np.random.RandomState(0).permutation(U).astype(np.int32)).to(dev)
rows_list = [perm[k * 512:(k + 1) * 512].contiguous() for k in range(U // 512)]
it2 = iter(rows_list * 20)
ms = timed(lambda: vae.step(next(it2), 0.2, 0.8, want_loss=True), 150, 20)
all_rows = torch.arange(U, dtype=torch.int32, device=dev)
def vae_eval():
    res = []
    for b0 in range(0, users.numel(), 8192):
        u = users[b0:b0 + 8192]
        S = vae.logits(u)
        E.mask_train(S, u, trc, cols=I)
        res.append(E.eval_scores(S, tec, [1, 2, 4, 3, 5], 20, users=u, cols=I))
    return torch.cat(res)
vae_eval(); torch.cuda.synchronize(); t0 = time.perf_counter(); r = vae_eval(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
# the same evaluation through the factor path ([g1 | 1]·[W_p1 | b_p1], pruned: no logits slab)
def vae_eval_factors():
    pf, qf = vae.eval_factors()
    return ev.evaluate_factors(pf, qf, users)
vae_eval_factors(); torch.cuda.synchronize(); t0 = time.perf_counter(); mf_ = vae_eval_factors(); torch.cuda.synchronize()
dtf = time.perf_counter() - t0
out["multivae"] = {"ms_per_step": ms, "users_per_sec_train": 512 / ms * 1e3, "batch": 512, "p_dim": [16, 32],
                   "eval_users_per_sec": users.numel() / dtf, "ndcg@10": float(mf_[2 * 20 + 9]),
                   "eval_users_per_sec_logits_slab": users.numel() / dt,
                   "ndcg@10_logits_slab": float(r.mean(0)[2 * 20 + 9].item()),
                   "note": "per-user inputs at evaluation (predict_accumulates_rows=False); a step writes the "
                           "[512][I] logits once and reads them three times (row statistics, dW_p1, dg1 — "
                           "the two gradients on the matrix cores, dLoss/dlogits never stored)"}
print(json.dumps(out))
