"""Power and shader clock while the bf16 filter, the int8 filter (and, for comparison, the fp32 loop and an idle GPU)
run back to back for ~2.5 s each: rocm-smi sampled from a side thread."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurec_amd import engine as E

dev = torch.device("cuda:0")
torch.manual_seed(1)
U, I, d = 16384, 40981, 64
P = (torch.randn(U, d, device=dev) * 0.01).contiguous()
Q = (torch.randn(I, d, device=dev) * 0.01).contiguous()
users = torch.arange(U, dtype=torch.int32, device=dev)
g, f = E.ScoreGemm(Q, U), E.ScoreFilter(Q, U)
f8 = E.ScoreFilter(Q, U, "int8")
n_tiles = 2 * ((I + 63) // 64)
M = torch.empty((U, (n_tiles + 3) // 4 * 4), dtype=torch.float32, device=dev)
eps = torch.empty(U, dtype=torch.float32, device=dev)


def fp32():
    E.call("nrhip_score_tilemax", E._ptr(P), P.stride(0), E._ptr(users), U, I, d, None, None, E._ptr(M), M.stride(0),
           E._ptr(g.ws), g.ws.numel(), E._stream())


def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True)
        out.append(r.stdout)
        time.sleep(0.25)


def run(name, fn, seconds=2.5):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        if fn is None:
            time.sleep(0.05)
        else:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
    stop.set(); th.join()
    print("==", name, "launches", n, "-> %.3f ms each" % ((time.time() - t0) * 1e3 / max(n, 1)))
    for o in out[1:4]:
        print(o.strip().replace("\n", " | ")[:400])


run("idle", None)
run("bf16 filter", lambda: f.tile_maxima(P, users, out=M, eps=eps))
run("int8 filter", lambda: f8.tile_maxima(P, users, out=M, eps=eps))
run("fp32 MFMA loop", fp32)
