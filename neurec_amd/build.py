"""Build libneurec_hip.so (hipcc, gfx950 only) next to this file.

    python -m neurec_amd.build [--force]

The library is built in-tree so that it travels with the source snapshot; it
is compiled for gfx950 and nothing else (no host fallback, no second arch).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "_obj")
LIB_PATH = os.path.join(HERE, "libneurec_hip.so")
STAMP = os.path.join(OBJ_DIR, "sources.sha256")

SOURCES = ["eval_select.hip", "score_gemm.hip", "sampler.hip", "spmm.hip", "bpr.hip", "adam.hip",
           "step.hip", "dense.hip", "vae.hip", "spmm_blocked.hip", "route.hip", "gemm.hip", "vae_wide.hip", "ngcf_wide.hip", "vae_fused.hip",
           "score_bf16.hip", "score_i8.hip", "eval_pipeline.hip"]
# per-file flags.  score_i8.hip: MFMA results in the unified VGPR file instead of AGPRs — its epilogue reads every
# accumulator of two sets per tile, and v_accvgpr_read per element doubled its VALU work (896 -> 78 reads in the loop)
FILE_FLAGS = {"score_i8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
# micro-benchmarks behind the design decisions in DESIGN.md (scripts/exp_*.py): their own library,
# nothing of it is linked into the product
EXP_SOURCES = ["experiments/gather_experiments.hip"]
EXP_LIB_PATH = os.path.join(HERE, "libneurec_exp.so")
HEADERS = ["nr_core.h", "nr_common.h"]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
         "-I", CSRC, "-I", os.path.join(ROOT, "include")] + os.environ.get("NEUREC_HIPCC_EXTRA", "").split()


def _existing_sources():
    return [s for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


def _digest():
    h = hashlib.sha256()
    for name in _existing_sources() + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    with open(os.path.join(ROOT, "include", "neurec_hip.h"), "rb") as f:
        h.update(f.read())
    # the flags without the checkout's absolute paths: the same tree digests the same wherever it lies
    h.update(" ".join(f.replace(ROOT, "<root>") for f in FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def is_current():
    if not (os.path.isfile(LIB_PATH) and os.path.isfile(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == _digest()


def build_extension(force=False, verbose=True):
    """Compile every .hip under csrc/ and link the shared library. Returns its path."""
    if not force and is_current():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _existing_sources()

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 and FILE_FLAGS.get(src):
            # the per-file flags are compiler-internal options: a hipcc that does not know one still builds the file
            # without it (same results, a slower kernel) instead of failing the whole library
            r2 = subprocess.run([HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
            if r2.returncode == 0:
                sys.stderr.write("neurec_amd.build: %s compiled WITHOUT %s (not accepted by this hipcc)\n"
                                 % (src, " ".join(FILE_FLAGS[src])))
                r = r2
        return src, obj, r

    objs = []
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        for src, obj, r in ex.map(compile_one, srcs):
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
            if verbose and r.stderr.strip():
                sys.stderr.write(r.stderr)
            objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(STAMP, "w") as f:
        f.write(_digest())
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


def build_experiments(verbose=True):
    """Compile csrc/experiments/ into libneurec_exp.so (used by scripts/exp_*.py only)."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    for src in EXP_SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.basename(src).replace(".hip", ".o"))
        r = subprocess.run([HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        objs.append(obj)
    # error reporting (nrhip_set_error) lives in the product library: link against it
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", EXP_LIB_PATH] + objs +
                       ["-L", HERE, "-lneurec_hip", "-Wl,-rpath,$ORIGIN"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", EXP_LIB_PATH)
    return EXP_LIB_PATH


if __name__ == "__main__":
    build_extension(force="--force" in sys.argv)
    if "--experiments" in sys.argv:
        build_experiments()
