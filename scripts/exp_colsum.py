"""colsum_stage1 takes 18-20 us for a 12 MB matrix inside the evaluation: the kernel, or where its input comes from?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurec_amd import engine as E
m = torch.rand(29858, 100, device="cuda")
def t(fn, n=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("colsum alone, back to back: %.1f us per call (two launches)" % t(lambda: E.colsum(m)))
def both():
    m.mul_(1.0000001)                 # a kernel that rewrites the matrix first (every XCD's L2 holds dirty lines of it)
    E.colsum(m)
print("rewrite + colsum: %.1f us; the rewrite alone: %.1f us" % (t(both), t(lambda: m.mul_(1.0000001))))
