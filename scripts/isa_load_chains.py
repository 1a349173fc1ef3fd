"""Which kernels wait for their loads one by one?  Compiles every csrc/*.hip to gfx950 assembly and counts, per kernel,
the vector loads, the `s_waitcnt vmcnt(0)` waits and the longest load -> wait-for-everything -> load chain.  A load
behind a branch (`if (c) x = p[i]`, `c ? p[i] : 0`) cannot be counted by the compiler's wait insertion: it waits for
ALL loads in flight before the next one is issued, so N such loads cost N memory round trips instead of one.  Found and
fixed this way in round 6: colsum_stage1 (64 in a row: 20 -> 6 us), gather_transpose_kernel (32: 13.5 -> 8.7 us), the BPR
heads' row and plan-key loads, the row loads of rank_compact_kernel / select_tiles_kernel.  A long chain is not always a
defect — pointer chasing (key -> id -> row) is a chain by nature — the table says where to look.
    python scripts/isa_load_chains.py [min_chain]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
min_chain = int(sys.argv[1]) if len(sys.argv) > 1 else 4
out_dir = tempfile.mkdtemp(prefix="isa_")
rows = []
for src in sorted(glob.glob(os.path.join(ROOT, "neurec_amd", "csrc", "*.hip"))):
    asm = os.path.join(out_dir, os.path.basename(src)[:-4] + ".s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "neurec_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", asm],
                       capture_output=True, text=True)
    if r.returncode != 0:
        print("(could not compile %s)" % os.path.basename(src))
        continue
    cur = None
    for ln in open(asm):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, loads, waits, chain, best, state = m.group(1), 0, 0, 0, 0, 0
            continue
        if cur is None:
            continue
        t = ln.strip()
        if t.startswith(("global_load", "buffer_load", "flat_load")):
            loads += 1
            if state == 2:
                chain += 1
                best = max(best, chain)
            else:
                chain = 0
            state = 1
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            waits += 1
            if state == 1:
                state = 2
        elif t.startswith("s_endpgm"):
            if best >= min_chain:
                try:
                    name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip() or cur
                except OSError:
                    name = cur
                rows.append((best, loads, waits, os.path.basename(src), name.replace("(anonymous namespace)::", "")[:110]))
            cur = None
for best, loads, waits, src, name in sorted(rows, reverse=True):
    print("chain %2d  loads %3d  vmcnt(0) waits %3d  %-18s %s" % (best, loads, waits, src, name))
