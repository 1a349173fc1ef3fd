"""The C-ABI shared library loads and exports every symbol include/neurec_hip.h declares
(no compute: this runs without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "neurec_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(nrhip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_whole_path():
    syms = declared_symbols()
    for must in ["nrhip_eval_scores", "nrhip_arg_topk", "nrhip_mask_train", "nrhip_score_gemm",
                 "nrhip_sample_bpr_epoch", "nrhip_randint_choice_batch", "nrhip_bpr_mf_grad",
                 "nrhip_adam_sparse_tf", "nrhip_adam_dense_tf", "nrhip_spmm_csr",
                 "nrhip_lightgcn_bpr_grad", "nrhip_last_error"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    from neurec_amd import build
    path = build.build_extension()            # no-op when current; cross-compiles without a GPU
    lib = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in neurec_hip.h but not exported: %s" % missing
    # 2: batch plans (ordered row-gradient sums) in the BPR heads; 3: one-launch BPR-MF step;
    # 4: nrhip_mf_steps with one loss reduction per call (per-step terms buffer)
    assert lib.nrhip_abi_version() == 4
    with open(os.path.join(ROOT, "include", "neurec_hip.h")) as f:
        assert "#define NRHIP_ABI_VERSION 4" in f.read()


def test_library_exports_nothing_undeclared():
    """The product library's C surface is exactly the header (experiments live in libneurec_exp.so);
    nrhip_set_error is the one internal hook (error text shared with that second library)."""
    import shutil
    import subprocess
    from neurec_amd import build
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", build.build_extension()], capture_output=True,
                         text=True, check=True).stdout
    exported = set(re.findall(r" T (nrhip_\w+)", out))
    assert exported - set(declared_symbols()) == {"nrhip_set_error"}


def test_python_binding_covers_the_header():
    from neurec_amd import _lib
    assert set(declared_symbols()) == set(_lib.EXPORTED)


def test_argument_errors_surface_as_python_exceptions():
    """Error convention of the boundary: status code + message -> ValueError/NotImplementedError
    (random_choice.pyx:23-37 / uni_evaluator.py:69 raise the same types)."""
    import ctypes as C
    from neurec_amd import _lib
    n = C.c_size_t(0)
    with pytest.raises(ValueError):
        _lib.call("nrhip_eval_workspace_bytes", 4, 0, C.byref(n))
    with pytest.raises(NotImplementedError):
        _lib.call("nrhip_score_gemm_workspace_bytes", 4, 100, 4096, C.byref(n))
    assert "4096" in _lib.last_error()
    _lib.call("nrhip_eval_workspace_bytes", 128, 20, C.byref(n))
    assert n.value > 0


def test_no_product_import_of_the_oracle():
    """The product must never route through oracle/ (or any CPU fallback)."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "neurec_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src:
                    bad.append(os.path.join(dirpath, fn))
    assert not bad, bad


def test_header_is_plain_c(tmp_path):
    """include/neurec_hip.h is the boundary a C / cgo / JNI binder compiles against: it must be valid C99
    on its own (no C++ in the signatures, every type it names declared by it or <stdint.h>/<stddef.h>)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "use_header.c"
    src.write_text('#include "neurec_hip.h"\n'
                   "int main(void) { nrhip_lightgcn_buffers a; nrhip_mf_buffers b; nrhip_ngcf_buffers c;\n"
                   "  (void)a; (void)b; (void)c; return NRHIP_ABI_VERSION == 3 ? 0 : 1; }\n")
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                          "-o", str(tmp_path / "use_header.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_a_library_built_from_other_sources_is_refused(monkeypatch):
    """A .so older than the kernels next to it must not be measured or tested by accident: the loader
    compares the build's source digest and raises (NEUREC_ALLOW_STALE_LIB=1 overrides)."""
    from neurec_amd import build
    build.build_extension()
    from neurec_amd import _lib
    assert not _lib.is_stale()
    monkeypatch.setattr(build, "is_current", lambda: False)
    monkeypatch.delenv("NEUREC_ALLOW_STALE_LIB", raising=False)
    with pytest.raises(ImportError, match="other sources"):
        _lib._load()
    monkeypatch.setenv("NEUREC_ALLOW_STALE_LIB", "1")
    assert _lib._load() is not None


def test_the_source_digest_does_not_depend_on_the_checkout_path(monkeypatch):
    from neurec_amd import build
    d0 = build._digest()
    monkeypatch.setattr(build, "FLAGS", [f.replace(build.ROOT, "/somewhere/else") for f in build.FLAGS])
    monkeypatch.setattr(build, "ROOT", "/somewhere/else")
    # (ROOT is also where include/neurec_hip.h is read from: keep the header readable)
    monkeypatch.setattr(build.os.path, "join", lambda *a, _j=build.os.path.join: _j(*a).replace("/somewhere/else", os.path.dirname(build.HERE)))
    assert build._digest() == d0


def test_the_filter_bound_constant_is_the_documented_formula():
    """kappa(d) of the bounded bf16 search (host code of the library, no GPU needed): 1.5 x (dropped terms 3.2 x 2^-18 +
    3 d' accumulations at 2^-23 + the fp32 chain's d' roundings at 2^-24), d' = d padded to the kernel's width; the
    workspace grows with rows, columns and width; widths beyond 128 are refused by name."""
    import ctypes as C
    from neurec_amd import _lib
    for d, dp in ((1, 16), (16, 16), (17, 32), (48, 48), (50, 64), (64, 64), (65, 128), (128, 128)):
        k = C.c_float(0)
        _lib.call("nrhip_score_filter_kappa", d, C.byref(k))
        want = np.float32(1.5) * (np.float32(3.2 * 2.0 ** -18) + np.float32(3.0 * dp) * np.float32(2.0 ** -23) +
                                  np.float32(dp) * np.float32(2.0 ** -24))
        assert abs(k.value - float(want)) <= 1e-6 * float(want), (d, k.value, float(want))
    sizes = []
    for rows, cols, d in ((1024, 5000, 64), (2048, 5000, 64), (2048, 50000, 64), (2048, 50000, 128)):
        n = C.c_size_t(0)
        _lib.call("nrhip_score_filter_workspace_bytes", rows, cols, d, C.byref(n))
        sizes.append(n.value)
    assert sizes == sorted(sizes) and sizes[0] > 0
    with pytest.raises(NotImplementedError, match="128"):
        _lib.call("nrhip_score_filter_kappa", 129, C.byref(C.c_float(0)))


def test_the_pruned_evaluation_workspace_serves_every_shorter_batch():
    """nrhip_eval_pruned sizes its level-2 workspace ONCE for batch_rows; a shorter last batch may
    take the tile-grouped rescoring the full batches were too large for (buckets beyond 1 GiB).  The size query is
    therefore monotone in rows — what it returns for the capacity covers every row count below it (host side; the
    kernels also fall back to the per-row rescoring when the buckets do not fit)."""
    import ctypes as C
    from neurec_amd import _lib
    q = lambda rows, cols: (_lib.call("nrhip_eval_tiles_bounded_workspace_bytes", rows, cols, 20, 23, C.byref(n)), n.value)[1]
    n = C.c_size_t(0)
    for cols in (40981, 262144, 300000, 393216, 1000000):
        sizes = [q(r, cols) for r in (1, 100, 2048, 8192, 16384, 20000, 32768, 65536)]
        assert all(a <= b for a, b in zip(sizes, sizes[1:])), (cols, sizes)
    # the case of the finding: ~300 k items at batch_rows 32,768 — its full batches take the packed buckets (r06), a
    # 20,000-row tail the strided ones: the capacity's workspace covers the tail's strided buckets (tiles x rows x 4 B)
    strided_tail = 2 * ((300000 + 63) // 64) * 20000 * 4
    assert q(32768, 300000) >= q(20000, 300000) > strided_tail
    # a million items: no strided form at all (31,250 tiles > the LDS histogram), the packed buckets are rows x n_keep
    assert q(8192, 1000000) - q(8192, 1000000 - 64) < (1 << 20)
