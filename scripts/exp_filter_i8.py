"""Level 1 of the pruned evaluation on the int8 matrix cores (csrc/score_i8.hip) against the bf16 filter and the fp32
MFMA loop, gowalla shape.  For each table kind: the largest |filter - fp32 chain| over all (user, tile) maxima relative
to the row's bound (must stay <= 1: the bound is derived), the two bounds side by side, the times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurec_amd import engine as E

dev = torch.device("cuda:0")
U, I = 29858, 40981
rows = 16384
users = torch.arange(rows, dtype=torch.int32, device=dev)


def hip_time(fn, n=10, w=3):
    for _ in range(w):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def tables(kind, d):
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    P = torch.randn(U, d, device=dev, generator=g)
    Q = torch.randn(I, d, device=dev, generator=g)
    if kind == "gauss":
        P, Q = P * 0.01, Q * 0.01
    elif kind == "heavy":                      # item norms over two decades, a few entries that dominate their row
        Q = Q * torch.exp(torch.randn(I, 1, device=dev, generator=g) * 1.2) * 0.02
        P = P * torch.exp(torch.randn(U, d, device=dev, generator=g) * 0.8) * 0.05
    elif kind == "spread":                     # exponents spread over 26 binades
        P = P * torch.exp2(torch.randint(-20, 7, (U, d), device=dev, generator=g).float())
        Q = Q * torch.exp2(torch.randint(-20, 7, (I, d), device=dev, generator=g).float())
    return P.contiguous(), Q.contiguous()


for d in (64, 32):
    for kind in ("gauss", "heavy", "spread"):
        P, Q = tables(kind, d)
        gm = E.ScoreGemm(Q, rows)
        n_tiles = 2 * ((I + 63) // 64)
        mld = (n_tiles + 3) // 4 * 4
        M0 = torch.empty((rows, mld), dtype=torch.float32, device=dev)

        def fp32():
            E.call("nrhip_score_tilemax", E._ptr(P), P.stride(0), E._ptr(users), rows, I, d, None, None, E._ptr(M0),
                   M0.stride(0), E._ptr(gm.ws), gm.ws.numel(), E._stream())
        fp32()
        line = "d=%d %-6s" % (d, kind)
        times = {}
        for arith in ("bf16", "int8"):
            f = E.ScoreFilter(Q, rows, arith)
            M1, eps = f.tile_maxima(P, users)
            torch.cuda.synchronize()
            A, B = M0[:, :n_tiles], M1[:, :n_tiles]
            fin = torch.isfinite(A)
            assert bool((torch.isfinite(B) == fin).all()), "pad tiles differ (%s)" % arith
            diff = torch.where(fin, (A - B).abs(), torch.zeros_like(A))
            ok = torch.isfinite(eps)
            ratio = (diff[ok] / eps[ok, None]).max().item() if bool(ok.any()) else float("nan")
            line += "  | %s: eps median %.3e  max diff/eps %.4f  unbounded rows %d" % (
                arith, eps[ok].median().item() if bool(ok.any()) else float("nan"), ratio, int((~ok).sum()))
            times[arith] = hip_time(lambda: f.tile_maxima(P, users, out=M1, eps=eps))
            times[arith + "_prep"] = hip_time(lambda: f.prepare(Q))
            del f
        t0 = hip_time(fp32)
        print(line)
        print("          fp32 loop %.3f ms   bf16 %.3f ms (items %.3f)   int8 %.3f ms (items %.3f)   int8 / bf16 = %.2f" % (
            t0, times["bf16"], times["bf16_prep"], times["int8"], times["int8_prep"], times["int8"] / times["bf16"]))
        del gm
