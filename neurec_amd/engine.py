"""Tensor-level front end of the HIP engine.

PyTorch-ROCm is used for device memory, streams and (elsewhere) collectives;
every computation below is a call into libneurec_hip.so through the C ABI
(include/neurec_hip.h) with raw device pointers.  Nothing here computes on the
CPU and nothing falls back to torch ops.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import call

NGCF_MAX_LAYERS = _lib.NGCF_MAX_LAYERS

METRIC_IDS = {"Precision": 1, "Recall": 2, "MAP": 3, "NDCG": 4, "MRR": 5}  # metric.h:111-117


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("neurec_amd needs a ROCm GPU (MI355X / gfx950); no CPU path exists")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype=None, allow_none=False):
    if t is None:
        if allow_none:
            return C.c_void_p(0)
        raise ValueError("tensor argument is None")
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("expected a CUDA/ROCm tensor, got %r" % (type(t),))
    if dtype is not None and t.dtype != dtype:
        raise TypeError("expected dtype %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


class Workspace:
    """A grow-only device scratch buffer (the C ABI never allocates)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes):
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=require_gpu())
        return self.buf


class DeviceCSR:
    """indptr (int64) + ascending indices (int32) of a sparse 0/1 matrix on the device."""

    def __init__(self, indptr, indices, n_cols):
        dev = require_gpu()
        if isinstance(indices, torch.Tensor):          # already on the device (synth.device_*): no host copy
            self.indptr = indptr.to(dev, torch.int64).contiguous()
            self.indices = indices.to(dev, torch.int32).contiguous()
            indptr = self.indptr.cpu().numpy()
        else:
            indptr = np.ascontiguousarray(indptr, dtype=np.int64)
            indices = np.ascontiguousarray(indices, dtype=np.int32)
            self.indptr = torch.from_numpy(indptr).to(dev)
            self.indices = torch.from_numpy(indices if len(indices) else np.zeros(1, np.int32)).to(dev)
        self.n_rows = len(indptr) - 1
        self.n_cols = int(n_cols)
        self.nnz = int(indptr[-1])
        self.h_indptr = indptr

    @staticmethod
    def from_scipy(mat):
        m = mat.tocsr().copy()
        m.sum_duplicates()
        m.sort_indices()
        return DeviceCSR(m.indptr, m.indices, m.shape[1])

    @staticmethod
    def from_dict(d, n_rows, n_cols):
        """{row: iterable of column ids} (the reference's user->items dicts)."""
        counts = np.zeros(n_rows + 1, dtype=np.int64)
        for r, items in d.items():
            counts[int(r) + 1] = len(items)
        indptr = np.cumsum(counts)
        indices = np.empty(int(indptr[-1]), dtype=np.int32)
        for r, items in d.items():
            b = indptr[int(r)]
            indices[b:b + len(items)] = np.sort(np.asarray(items, dtype=np.int32))
        return DeviceCSR(indptr, indices, n_cols)

    def rows(self, lo, hi):
        """rows [lo, hi) as a CSR of their own (row r = row lo + r here; the columns stay): a rank's users of a
        row-sharded evaluation (sharded.ShardedEvaluator)"""
        lo, hi = int(lo), int(hi)
        b, e = int(self.h_indptr[lo]), int(self.h_indptr[hi])
        idx = self.indices[b:e] if e > b else torch.zeros(1, dtype=torch.int32, device=self.indices.device)
        return DeviceCSR(self.indptr[lo:hi + 1] - b, idx, self.n_cols)

    def check_sorted(self, n_cols=None):
        """Raises ValueError unless every row's indices ascend (a repeated id is tolerated: repeats are neighbours then)
        and lie in [0, n_cols): what the planned strikes of the pruned evaluation assume of a train matrix (one device
        pass, one host read)."""
        n_cols = self.n_cols if n_cols is None else int(n_cols)
        if self.nnz == 0:
            return
        idx = self.indices[:self.nnz].long()
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= n_cols:
            raise ValueError("CSR column ids span [%d, %d], outside [0, %d)" % (lo, hi, n_cols))
        if self.nnz > 1:
            key = self.row_of().long() * n_cols + idx
            if bool((key[1:] < key[:-1]).any()):
                raise ValueError("CSR rows must hold ascending column ids (sort the rows: DeviceCSR.from_scipy does)")

    def row_of(self):
        """User id of every CSR position (users_list of data/sampler.py:24-39)."""
        if self.nnz == 0:
            return torch.zeros(1, dtype=torch.int32, device=self.indptr.device)
        return torch.repeat_interleave(torch.arange(self.n_rows, dtype=torch.int32, device=self.indptr.device),
                                       self.indptr[1:] - self.indptr[:-1], output_size=self.nnz)


# ----------------------------------------------------------------------------- evaluator
_eval_ws = Workspace()
_gemm_ws = Workspace()
_misc_ws = Workspace()


def mask_train(scores, users, train_csr, cols=None):
    """scores[r, train items of users[r]] = -inf in place (uni_evaluator.py:140-143)."""
    if scores.dim() != 2 or scores.stride(1) != 1 or scores.dtype != torch.float32:
        raise ValueError("scores must be a 2-D float32 tensor with unit inner stride")
    rows = scores.shape[0]
    cols = scores.shape[1] if cols is None else cols
    call("nrhip_mask_train", C.c_void_p(scores.data_ptr()), scores.stride(0),
         _ptr(users, torch.int32, allow_none=True), rows, cols, _ptr(train_csr.indptr),
         _ptr(train_csr.indices), _stream())


def eval_scores(scores, truth_csr, metric_ids, top_k, users=None, cols=None, out=None,
                want_topk=False, want_exact_count=False):
    """Top-K + metrics of each score row; returns float32 [rows, len(metric_ids)*top_k]."""
    if scores.dim() != 2 or scores.stride(1) != 1:
        raise ValueError("scores must be 2-D with unit inner stride")
    rows = scores.shape[0]
    cols = scores.shape[1] if cols is None else cols
    dev = scores.device
    nm = len(metric_ids)
    if out is None:
        out = torch.empty((rows, nm * top_k), dtype=torch.float32, device=dev)
    topk = torch.empty((rows, top_k), dtype=torch.int32, device=dev) if want_topk else None
    nex = torch.zeros(1, dtype=torch.int32, device=dev) if want_exact_count else None
    nbytes = C.c_size_t(0)
    ids = (C.c_int * nm)(*[int(m) for m in metric_ids])
    if top_k > 128:                 # beyond the parallel selection's cut-off: the sequential replay, any K
        call("nrhip_eval_any_k_workspace_bytes", rows, cols, top_k, C.byref(nbytes))
        ws = _eval_ws.get(nbytes.value)
        call("nrhip_eval_scores_any_k", C.c_void_p(scores.data_ptr()), scores.stride(0), rows, cols,
             _ptr(users, torch.int32, allow_none=True), _ptr(truth_csr.indptr), _ptr(truth_csr.indices), ids, nm,
             top_k, _ptr(out, torch.float32), _ptr(topk, allow_none=True), _ptr(ws), ws.numel(), _stream())
        if nex is not None:
            nex.fill_(rows)          # every row took the exact replay
        res = [out] + ([topk] if want_topk else []) + ([nex] if want_exact_count else [])
        return res[0] if len(res) == 1 else tuple(res)
    call("nrhip_eval_workspace_bytes", rows, top_k, C.byref(nbytes))
    ws = _eval_ws.get(nbytes.value)
    call("nrhip_eval_scores", C.c_void_p(scores.data_ptr()), scores.stride(0), rows, cols,
         _ptr(users, torch.int32, allow_none=True), _ptr(truth_csr.indptr),
         _ptr(truth_csr.indices), ids, nm, top_k, _ptr(out, torch.float32),
         _ptr(topk, allow_none=True), _ptr(nex, allow_none=True), _ptr(ws), ws.numel(), _stream())
    res = [out]
    if want_topk:
        res.append(topk)
    if want_exact_count:
        res.append(nex)
    return res[0] if len(res) == 1 else tuple(res)


def arg_topk(scores, top_k, cols=None, want_exact_count=False):
    """Per-row arg-top-K (util/cython/arg_topk.pyx:16-35) -> int32 [rows, top_k]."""
    rows = scores.shape[0]
    cols = scores.shape[1] if cols is None else cols
    out = torch.empty((rows, top_k), dtype=torch.int32, device=scores.device)
    nex = torch.zeros(1, dtype=torch.int32, device=scores.device) if want_exact_count else None
    nbytes = C.c_size_t(0)
    call("nrhip_eval_workspace_bytes", rows, min(top_k, 128), C.byref(nbytes))
    ws = _eval_ws.get(nbytes.value)
    call("nrhip_arg_topk", C.c_void_p(scores.data_ptr()), scores.stride(0), rows, cols, top_k,
         _ptr(out), _ptr(nex, allow_none=True), _ptr(ws), ws.numel(), _stream())
    return (out, nex) if want_exact_count else out


def colsum(mat):
    """fp64 column sums of a float32 matrix (deterministic order)."""
    rows, cols = mat.shape
    out = torch.empty(cols, dtype=torch.float64, device=mat.device)
    nbytes = C.c_size_t(0)
    call("nrhip_colsum_workspace_bytes", rows, cols, C.byref(nbytes))
    ws = _misc_ws.get(nbytes.value)
    call("nrhip_colsum_f64", _ptr(mat, torch.float32), mat.stride(0), rows, cols, _ptr(out),
         _ptr(ws), ws.numel(), _stream())
    return out


class ScoreGemm:
    """S = P[users] @ Q.T on the fp32 matrix cores; Q is prepared once per evaluation."""

    def __init__(self, item_table, max_rows):
        self.cols, self.d = item_table.shape
        self.max_rows = int(max_rows)
        nbytes = C.c_size_t(0)
        call("nrhip_score_gemm_workspace_bytes", self.max_rows, self.cols, self.d, C.byref(nbytes))
        self.ws = torch.empty(nbytes.value, dtype=torch.uint8, device=item_table.device)
        self.ld = (self.cols + 63) // 64 * 64
        self.prepare(item_table)

    def prepare(self, item_table):
        """(Re)load the item side — call once per evaluation, after the table changed."""
        if tuple(item_table.shape) != (self.cols, self.d):
            raise ValueError("item table shape changed")
        call("nrhip_score_gemm_prepare_items", _ptr(item_table, torch.float32),
             item_table.stride(0), self.cols, self.d, _ptr(self.ws), self.ws.numel(), _stream())

    def new_score_buffer(self, rows=None):
        rows = self.max_rows if rows is None else rows
        return torch.empty((rows, self.ld), dtype=torch.float32, device=self.ws.device)

    def __call__(self, user_table, users, out=None):
        rows = user_table.shape[0] if users is None else users.numel()
        if rows > self.max_rows:
            raise ValueError("batch of %d rows > prepared max_rows=%d" % (rows, self.max_rows))
        if out is None:
            out = self.new_score_buffer(rows)
        call("nrhip_score_gemm", _ptr(user_table, torch.float32), user_table.stride(0),
             _ptr(users, torch.int32, allow_none=True), rows, self.cols, self.d,
             C.c_void_p(out.data_ptr()), out.stride(0), _ptr(self.ws), self.ws.numel(), _stream())
        return out[:rows]


    def tile_maxima(self, user_table, users, train_csr, out=None, plan=None, row_of=None, row_lo=0, filt=None):
        """Pruned evaluation, level 1: M[r][t] = max admissible score of user row r over 32-item
        tile t (train items and pad columns excluded); the scores themselves are never stored.
        With a TileStrikePlan of the train matrix: the scoring loop runs without the train lists and the
        planned (user, tile) pairs are recomputed with their strikes afterwards — same M, bit for bit.
        `row_of` [n_users] int32 maps a user to its row in the whole evaluation order (None: row = user) and
        `row_lo` is the first row of this batch in that order.
        With `filt` (a ScoreFilter of the same item table; needs `plan`) the scoring loop is the bf16 bounded filter
        and the return value is (M, eps): |M[r][t] - the fp32 chain's maximum| <= eps[r] (the planned pairs are
        recomputed in fp32: error 0)."""
        rows = user_table.shape[0] if users is None else users.numel()
        if rows > self.max_rows:
            raise ValueError("batch of %d rows > prepared max_rows=%d" % (rows, self.max_rows))
        n_tiles = 2 * ((self.cols + 63) // 64)          # 32-item tiles
        mld = (n_tiles + 3) // 4 * 4
        if out is None:
            out = torch.empty((rows, mld), dtype=torch.float32, device=self.ws.device)
        eps = None
        if filt is not None:
            if plan is None:
                raise ValueError("the bounded filter leaves the train items to the planned fix-up: pass the plan")
            _, eps = filt.tile_maxima(user_table, users, out=out)
        else:
            call("nrhip_score_tilemax", _ptr(user_table, torch.float32), user_table.stride(0),
                 _ptr(users, torch.int32, allow_none=True), rows, self.cols, self.d,
                 None if plan is not None else _ptr(train_csr.indptr),
                 None if plan is not None else _ptr(train_csr.indices), _ptr(out, torch.float32),
                 out.stride(0), _ptr(self.ws), self.ws.numel(), _stream())
        if plan is not None:
            if plan.cols != self.cols:
                raise ValueError("strike plan built for %d items, scoring %d" % (plan.cols, self.cols))
            if callable(row_of):              # built while the scoring loop already runs (the fix-up alone reads it)
                row_of = row_of()
            call("nrhip_score_tilemax_fix", _ptr(user_table, torch.float32), user_table.stride(0), self.d, self.cols,
                 _ptr(plan.chunk_tile, torch.int32), _ptr(plan.chunk_begin, torch.int64), plan.n_chunks,
                 _ptr(plan.tile_ptr, torch.int64), _ptr(plan.user, torch.int32), _ptr(plan.mask, torch.int32),
                 _ptr(row_of, torch.int32, allow_none=True), int(row_lo), rows, _ptr(out, torch.float32),
                 out.stride(0), _ptr(self.ws), self.ws.numel(), _stream())
        return out[:rows] if filt is None else (out[:rows], eps)


class ScoreFilter:
    """Level 1 of the pruned evaluation as a bounded filter on the matrix cores: tile maxima, each within eps[row]
    of the fp32 chain's value.  arith="bf16" (csrc/score_bf16.hip, d <= 128): a three-term bf16 expansion of the fp32
    products; arith="int8" (csrc/score_i8.hip, d <= 128): 15-bit fixed point in exact integer accumulators, the bound
    derived from the quantisation (eps = NaN for rows it cannot bound).  Nothing here is ranked:
    nrhip_eval_tiles_bounded rescores the chosen tiles with the fp32 chain and accepts a row only if its bound
    certifies the choice."""

    ARITH = {"bf16": ("nrhip_score_filter_", 128, 1), "int8": ("nrhip_score_filter_i8_", 128, 2)}

    @staticmethod
    def supports(d, arith="bf16"):
        return int(d) <= ScoreFilter.ARITH[arith][1]

    def __init__(self, item_table, max_rows, arith="bf16"):
        if arith not in self.ARITH:
            raise ValueError("arith must be one of %s, got %r" % (sorted(self.ARITH), arith))
        self.arith = arith
        self._prefix, _, self.use_filter = self.ARITH[arith]      # use_filter: nrhip_eval_pruned's selector
        self.cols, self.d = item_table.shape
        self.max_rows = int(max_rows)
        nbytes = C.c_size_t(0)
        call(self._prefix + "workspace_bytes", self.max_rows, self.cols, self.d, C.byref(nbytes))
        self.ws = torch.empty(nbytes.value, dtype=torch.uint8, device=item_table.device)
        self.kappa = None
        if arith == "bf16":
            k = C.c_float(0)
            call("nrhip_score_filter_kappa", self.d, C.byref(k))
            self.kappa = float(k.value)
        self.prepare(item_table)

    def prepare(self, item_table):
        if tuple(item_table.shape) != (self.cols, self.d):
            raise ValueError("item table %s, prepared for %s" % (tuple(item_table.shape), (self.cols, self.d)))
        if item_table.stride(1) != 1:
            item_table = item_table.contiguous()
        call(self._prefix + "prepare_items", C.c_void_p(item_table.data_ptr()), item_table.stride(0), self.cols,
             self.d, _ptr(self.ws), self.ws.numel(), self.max_rows, _stream())

    def tile_maxima(self, user_table, users, out=None, eps=None):
        """(M, eps): M[r][t] ~ max score of row r over 32-item tile t (pad columns excluded, train items NOT struck),
        |M[r][t] - fp32 chain maximum| <= eps[r]."""
        rows = user_table.shape[0] if users is None else users.numel()
        if rows > self.max_rows:
            raise ValueError("batch of %d rows > prepared max_rows=%d" % (rows, self.max_rows))
        n_tiles = 2 * ((self.cols + 63) // 64)
        mld = (n_tiles + 3) // 4 * 4
        if out is None:
            out = torch.empty((rows, mld), dtype=torch.float32, device=self.ws.device)
        if eps is None:
            eps = torch.empty(rows, dtype=torch.float32, device=self.ws.device)
        call(self._prefix + "tilemax", _ptr(user_table, torch.float32), user_table.stride(0),
             _ptr(users, torch.int32, allow_none=True), rows, self.cols, self.d, _ptr(out, torch.float32),
             out.stride(0), _ptr(eps, torch.float32), _ptr(self.ws), self.ws.numel(), self.max_rows, _stream())
        return out[:rows], eps[:rows]


class ScoreGemmWide:
    """ScoreGemm's interface for factor widths beyond the scoring loop's 128 (NGCF at the paper's 64 / [64, 64, 64]
    concatenates to 256 columns): S = P[users] @ Q.T through the general fp32-MFMA GEMM (csrc/gemm.hip), both
    tables read as they lie — the same k-ascending fmaf chain per score.  The evaluation then takes the materialised
    path (score slab -> train mask -> select); there is no tile-maxima form at these widths."""

    wide = True

    def __init__(self, item_table, max_rows):
        self.cols, self.d = item_table.shape
        self.max_rows = int(max_rows)
        dev = item_table.device
        self.ld = (self.cols + 63) // 64 * 64
        self.Pg = torch.empty((self.max_rows, self.d), dtype=torch.float32, device=dev)
        self.prepare(item_table)

    def prepare(self, item_table):
        """Both factor tables are read as they lie (rows = the contraction index contiguous): nothing to copy."""
        if tuple(item_table.shape) != (self.cols, self.d):
            raise ValueError("item table shape changed")
        self.Q = item_table

    def new_score_buffer(self, rows=None):
        rows = self.max_rows if rows is None else rows
        return torch.empty((rows, self.ld), dtype=torch.float32, device=self.Q.device)

    def __call__(self, user_table, users, out=None):
        rows = user_table.shape[0] if users is None else users.numel()
        if rows > self.max_rows:
            raise ValueError("batch of %d rows > prepared max_rows=%d" % (rows, self.max_rows))
        if out is None:
            out = self.new_score_buffer(rows)
        if rows == 0:
            return out[:0]
        src, ld = user_table, user_table.stride(0)
        if users is not None:
            rows_gather(users, user_table, self.Pg[:rows])
            src, ld = self.Pg, self.d
        call("nrhip_gemm_f32", _ptr(src, torch.float32), ld, 1, _ptr(self.Q, torch.float32), self.Q.stride(0), 1, rows,
             self.cols, self.d, C.c_void_p(out.data_ptr()), out.stride(0), 0, None, -1, 1, None, 0, _stream())
        return out[:rows]


def score_gemm_for(item_table, max_rows):
    """The scoring object for an item factor table: the operand-swizzled scoring loop up to 128 columns, the general
    GEMM beyond."""
    return (ScoreGemmWide if item_table.shape[1] > 128 else ScoreGemm)(item_table, max_rows)


class TileStrikePlan:
    """Which (user, 32-item tile) pairs hold a train item, and which items of the tile — built ONCE per train
    matrix by three native launches (nrhip_tile_strike_plan: head count per tile, layout, fill; r05 — it was
    construction-time torch unique / bincount / cumsum).  Grouped by tile, cut into chunks of <= 32 pairs of one
    tile, the unit one wave of tilemax_fix_kernel recomputes (csrc/score_gemm.hip); the order of the pairs inside a
    tile is unspecified (every pair is independent in the fix-up pass).
    uni_evaluator.py:132-140 strikes ranking_score[u][train items of u] = -inf on the host for every batch."""

    def __init__(self, train_csr, cols=None):
        dev = train_csr.indptr.device
        self.cols = int(train_csr.n_cols if cols is None else cols)
        # the head-of-pair rule of the native build needs ascending rows, its histograms ids below cols
        train_csr.check_sorted(self.cols)
        U, nnz = int(train_csr.n_rows), int(train_csr.nnz)
        n_tiles = 2 * ((self.cols + 63) // 64)
        i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
        i64 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int64, device=dev)
        user, mask = i32(nnz), i32(nnz)
        self.tile_ptr = i64(n_tiles + 1)
        cap = nnz // 32 + n_tiles + 1
        chunk_tile, chunk_begin = i32(cap), i64(cap)
        counts, ws = i32(2), i32(2 * n_tiles)
        call("nrhip_tile_strike_plan", _ptr(train_csr.indptr, torch.int64), _ptr(train_csr.indices, torch.int32), U,
             self.cols, _ptr(user), _ptr(mask), _ptr(self.tile_ptr), _ptr(chunk_tile), _ptr(chunk_begin), _ptr(counts),
             _ptr(ws), ws.numel() * 4, _stream())
        self.n_pairs, self.n_chunks = (int(x) for x in counts.cpu())      # the one host round trip of the build
        self.user, self.mask = user[:max(self.n_pairs, 1)], mask[:max(self.n_pairs, 1)]
        self.chunk_tile, self.chunk_begin = chunk_tile[:max(self.n_chunks, 1)], chunk_begin[:max(self.n_chunks, 1)]


_tiles_ws = Workspace()


def eval_tiles(M, user_table, gemm, users, train_csr, truth_csr, metric_ids, top_k, out, flags, eps=None,
               n_keep=None, grouped=True):
    """Pruned evaluation, level 2 (nrhip_eval_tiles): metric rows into `out`, tie flags into
    `flags` (int32 per row; flagged rows must be recomputed from full score rows).  `gemm` is the
    ScoreGemm whose prepared (k-major) item copy the rescoring reads.  With `eps` (ScoreFilter.tile_maxima) the maxima
    are bounded, not exact: `n_keep` tiles are rescored and a row stands only if its bound certifies the choice
    (nrhip_eval_tiles_bounded).  `grouped`: the rescoring bucketed by tile (same scores; False = nrhip_eval_tiles'
    per-row kernel, exact maxima only)."""
    rows = M.shape[0]
    nbytes = C.c_size_t(0)
    nm = len(metric_ids)
    ids = (C.c_int * nm)(*[int(m) for m in metric_ids])
    if eps is not None or grouped:
        n_keep = int(top_k + 1 if n_keep is None or eps is None else n_keep)
        call("nrhip_eval_tiles_bounded_workspace_bytes", rows, gemm.cols, top_k, n_keep, C.byref(nbytes))
        ws = _tiles_ws.get(nbytes.value)
        call("nrhip_eval_tiles_bounded", C.c_void_p(M.data_ptr()), M.stride(0),
             _ptr(eps, torch.float32, allow_none=True), n_keep,
             _ptr(user_table, torch.float32), user_table.stride(0), _ptr(gemm.ws), gemm.d,
             _ptr(users, torch.int32, allow_none=True), rows, gemm.cols,
             _ptr(train_csr.indptr), _ptr(train_csr.indices), _ptr(truth_csr.indptr),
             _ptr(truth_csr.indices), ids, nm, top_k, _ptr(out, torch.float32),
             _ptr(flags, torch.int32), _ptr(ws), ws.numel(), _stream())
        return out
    call("nrhip_eval_tiles_workspace_bytes", rows, top_k, C.byref(nbytes))
    ws = _tiles_ws.get(nbytes.value)
    call("nrhip_eval_tiles", C.c_void_p(M.data_ptr()), M.stride(0), _ptr(user_table, torch.float32),
         user_table.stride(0), _ptr(gemm.ws), gemm.d,
         _ptr(users, torch.int32, allow_none=True), rows, gemm.cols,
         _ptr(train_csr.indptr), _ptr(train_csr.indices), _ptr(truth_csr.indptr),
         _ptr(truth_csr.indices), ids, nm, top_k, _ptr(out, torch.float32),
         _ptr(flags, torch.int32), _ptr(ws), ws.numel(), _stream())
    return out


class PrunedEvaluation:
    """nrhip_eval_pruned: the pruned evaluation of a whole user list in one native call (tile search, planned train
    strikes, fp32 rescoring + ranking + certificate + metrics per batch, then the column sums and the flagged-row
    count).  Holds the per-evaluator buffers; `run` returns (per_user [n][M*K] float32, flags [n] int32,
    sums [M*K + 2] float64 — the last two entries are the number of flagged rows and, among them, of rows whose
    certificate failed)."""

    def __init__(self, gemm, filt, plan, train_csr, truth_csr, metric_ids, top_k, n_keep, batch_rows):
        from ._lib import EvalPrunedArgs
        self.gemm, self.filt, self.plan = gemm, filt, plan
        self.train, self.truth = train_csr, truth_csr
        self.top_k, self.n_keep, self.batch_rows = int(top_k), int(n_keep), int(batch_rows)
        self.nm = len(metric_ids)
        self.ids = (C.c_int * self.nm)(*[int(m) for m in metric_ids])
        dev = gemm.ws.device
        n_tiles = 2 * ((gemm.cols + 63) // 64)
        self.mld = (n_tiles + 3) // 4 * 4
        self.M = torch.empty((self.batch_rows, self.mld), dtype=torch.float32, device=dev)
        self.eps = torch.empty(self.batch_rows, dtype=torch.float32, device=dev)
        nb = C.c_size_t(0)
        call("nrhip_eval_tiles_bounded_workspace_bytes", self.batch_rows, gemm.cols, self.top_k, self.n_keep, C.byref(nb))
        self.tiles_ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        self.sums = torch.empty(self.nm * self.top_k + 2, dtype=torch.float64, device=dev)
        self.cs_ws = None
        self.args = EvalPrunedArgs()

    def run(self, user_table, item_table, users, row_of, per_user, flags, prepare_items=True):
        a, g, f, pl = self.args, self.gemm, self.filt, self.plan
        n = users.numel()
        # the argument block of the previous call stands when the six tensors lie where they lay (the block holds nothing
        # of them but pointers, strides and counts; everything else in it belongs to this object): an evaluation per
        # epoch hands over the same tables, user list and buffers every time, and filling ~40 fields costs more than a
        # small launch
        key = (user_table.data_ptr(), user_table.stride(0), item_table.data_ptr(), item_table.stride(0), item_table.stride(1),
               users.data_ptr(), n, row_of.data_ptr(), per_user.data_ptr(), flags.data_ptr(), user_table.dtype,
               item_table.dtype, users.dtype, per_user.dtype, flags.dtype, row_of.dtype, user_table.stride(1),
               per_user.is_contiguous())
        if getattr(self, "_last", None) == key:
            a.prepare_items = int(prepare_items)
            call("nrhip_eval_pruned", C.byref(a), _stream())
            return per_user, flags, self.sums
        if getattr(self, "_cs_n", None) != n:
            nb = C.c_size_t(0)
            call("nrhip_colsum_workspace_bytes", max(n, 1), self.nm * self.top_k, C.byref(nb))
            if self.cs_ws is None or self.cs_ws.numel() < nb.value:
                self.cs_ws = torch.empty(max(nb.value, 256), dtype=torch.uint8, device=g.ws.device)
            self._cs_n = n
        if item_table.stride(1) != 1:
            item_table = item_table.contiguous()
            key = None                                       # (a temporary copy: nothing to remember)
        ptr = lambda t: t.data_ptr()
        a.P, a.ldp, a.Q, a.ldq, a.d, a.cols = ptr(user_table), user_table.stride(0), ptr(item_table), item_table.stride(0), g.d, g.cols
        a.users, a.n_users, a.batch_rows = ptr(users), n, self.batch_rows
        a.tr_indptr, a.tr_indices = ptr(self.train.indptr), ptr(self.train.indices)
        a.truth_indptr, a.truth_indices = ptr(self.truth.indptr), ptr(self.truth.indices)
        a.chunk_tile, a.chunk_begin, a.n_chunks = ptr(pl.chunk_tile), ptr(pl.chunk_begin), pl.n_chunks
        a.tile_ptr, a.plan_user, a.plan_mask, a.row_of = ptr(pl.tile_ptr), ptr(pl.user), ptr(pl.mask), ptr(row_of)
        a.metric_ids, a.n_metric, a.top_k, a.n_keep = self.ids, self.nm, self.top_k, self.n_keep
        # prepare_items: False / True, or 2 = the item side without the fp32 scoring loop's operand copy (filter only)
        a.use_filter, a.prepare_items = (f.use_filter if f is not None else 0), int(prepare_items)
        a.gemm_ws, a.gemm_ws_bytes = ptr(g.ws), g.ws.numel()
        a.filter_ws, a.filter_ws_bytes = (ptr(f.ws), f.ws.numel()) if f is not None else (None, 0)
        a.tiles_ws, a.tiles_ws_bytes = ptr(self.tiles_ws), self.tiles_ws.numel()
        a.M, a.mld, a.eps = ptr(self.M), self.mld, ptr(self.eps)
        a.out, a.flags, a.sums = ptr(per_user), ptr(flags), ptr(self.sums)
        a.colsum_ws, a.colsum_ws_bytes = ptr(self.cs_ws), self.cs_ws.numel()
        if user_table.dtype != torch.float32 or item_table.dtype != torch.float32 or users.dtype != torch.int32 or \
                per_user.dtype != torch.float32 or flags.dtype != torch.int32 or row_of.dtype != torch.int32:
            raise TypeError("nrhip_eval_pruned: float32 tables / output, int32 users / flags / row table")
        if not (user_table.stride(1) == 1 and users.is_contiguous() and per_user.is_contiguous() and flags.is_contiguous()):
            raise ValueError("nrhip_eval_pruned: contiguous arguments")
        call("nrhip_eval_pruned", C.byref(a), _stream())
        self._last = key
        return per_user, flags, self.sums

    def redo(self, n_flagged, slab, reload_items):
        """nrhip_eval_redo behind run(): the n_flagged rows run() flagged (the count it left behind the sums, read by
        the caller) ranked again from full fp32 score rows in `slab` [rows][ld] and written over their rows of run()'s
        per_user; the sums retaken.  reload_items: the scoring engine's item side first — 2: run() ran with
        prepare_items = 2 on this table (only the operand copy is missing), 1: all of it.  Returns the sums tensor."""
        from ._lib import EvalRedoArgs
        n_flagged = int(n_flagged)
        rows = min(int(slab.shape[0]), self.batch_rows)
        dev = slab.device
        r = getattr(self, "_redo_args", None)
        if r is None:
            r = self._redo_args = EvalRedoArgs()
            self._redo_count = torch.zeros(1, dtype=torch.int32, device=dev)
            self._redo_rows = self._redo_fixed = self._redo_ws = None
        if self._redo_rows is None or self._redo_rows.shape[1] < n_flagged:
            self._redo_rows = torch.empty((2, max(n_flagged, 256)), dtype=torch.int32, device=dev)
        if self._redo_fixed is None or self._redo_fixed.shape[0] < rows:
            self._redo_fixed = torch.empty((rows, self.nm * self.top_k), dtype=torch.float32, device=dev)
            nb = C.c_size_t(0)
            call("nrhip_eval_workspace_bytes", rows, self.top_k, C.byref(nb))
            self._redo_ws = torch.empty(max(nb.value, 256), dtype=torch.uint8, device=dev)
        if slab.dtype != torch.float32 or slab.stride(1) != 1:
            raise TypeError("nrhip_eval_redo: a float32 score slab with contiguous rows")
        r.ev = C.pointer(self.args)
        r.n_flagged, r.reload_items = n_flagged, int(reload_items)
        r.scores, r.lds, r.slab_rows = slab.data_ptr(), slab.stride(0), rows
        r.rows, r.row_users = self._redo_rows[0].data_ptr(), self._redo_rows[1].data_ptr()
        r.count, r.fixed = self._redo_count.data_ptr(), self._redo_fixed.data_ptr()
        r.ws, r.ws_bytes = self._redo_ws.data_ptr(), self._redo_ws.numel()
        call("nrhip_eval_redo", C.byref(r), _stream())
        return self.sums


# ----------------------------------------------------------------------------- sampler
def sample_bpr_epoch(train_csr, row_of, n_items, neg_num, seed, epoch, shuffle=True, begin=0,
                     count=None, out=None):
    """One epoch (or a slice of it) of BPR triplets: (users, pos, neg) int32 device tensors."""
    n_inter = train_csr.nnz
    count = n_inter - begin if count is None else count
    dev = train_csr.indptr.device
    if out is None:
        users = torch.empty(max(count, 1), dtype=torch.int32, device=dev)
        pos = torch.empty(max(count, 1), dtype=torch.int32, device=dev)
        neg = torch.empty(max(count * neg_num, 1), dtype=torch.int32, device=dev)
    else:
        users, pos, neg = out
    call("nrhip_sample_bpr_epoch", _ptr(train_csr.indptr), _ptr(train_csr.indices),
         _ptr(row_of, torch.int32), n_inter, n_items, neg_num, C.c_uint64(seed & (2**64 - 1)),
         C.c_uint64(epoch), 1 if shuffle else 0, begin, count, _ptr(users), _ptr(pos), _ptr(neg),
         _stream())
    return users[:count], pos[:count], neg[:count * neg_num]


def sample_instances_epoch(rows, n_items, neg_num, pointwise, seed, epoch, shuffle, begin, count, out):
    """One epoch (or a slice) of the instance stream of InstanceRows `rows` (nrhip_sample_instances_epoch):
    out = (users, recent-or-None, items, neg-or-None, labels-or-None) device buffers, filled in place."""
    users, recent, items, neg, labels = out
    call("nrhip_sample_instances_epoch", _ptr(rows.seq_ptr), _ptr(rows.seq), _ptr(rows.excl_ptr), _ptr(rows.excl),
         _ptr(rows.inst_ptr), _ptr(rows.inst_row), _ptr(rows.row_user), rows.n_inst, rows.high_order, n_items,
         neg_num, 1 if pointwise else 0, C.c_uint64(seed & (2**64 - 1)), C.c_uint64(epoch), 1 if shuffle else 0,
         begin, count, _ptr(users), _ptr(recent, allow_none=True), _ptr(items), _ptr(neg, allow_none=True),
         _ptr(labels, torch.float32, allow_none=True), _stream())


def randint_choice_batch(high, sizes, exclusion_csr, replace, seed, counter):
    """Device half of batch_randint_choice: returns an int32 tensor of sum(sizes) draws."""
    dev = require_gpu()
    sizes = np.asarray(sizes, dtype=np.int64)
    off = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    total = int(off[-1])
    d_off = torch.from_numpy(off).to(dev)
    out = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    eptr = exclusion_csr.indptr if exclusion_csr is not None else None
    eidx = exclusion_csr.indices if exclusion_csr is not None else None
    call("nrhip_randint_choice_batch", int(high), len(sizes), total, _ptr(d_off),
         _ptr(eptr, allow_none=True), _ptr(eidx, allow_none=True), 1 if replace else 0,
         C.c_uint64(seed & (2**64 - 1)), C.c_uint64(counter), _ptr(out), _stream())
    return out[:total], off


# ----------------------------------------------------------------------------- training
class AdamState:
    """fp32 running powers of beta1/beta2, advanced like TF-1.12's _finish."""

    def __init__(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr = np.float32(lr)
        self.beta1, self.beta2, self.eps = np.float32(beta1), np.float32(beta2), np.float32(eps)
        self.b1p, self.b2p = np.float32(beta1), np.float32(beta2)
        self.t = 0

    def alpha(self):
        one = np.float32(1.0)
        return np.float32(self.lr * np.sqrt(one - self.b2p) / (one - self.b1p))

    def advance(self):
        self.b1p = np.float32(self.b1p * self.beta1)
        self.b2p = np.float32(self.b2p * self.beta2)
        self.t += 1

    def alpha_table(self, n_steps):
        """alpha of steps 1..n_steps (index 0 unused) with the same fp32 operations as alpha() /
        advance(): running products by cumprod (sequential fp32 multiplications), then the formula."""
        one = np.float32(1.0)
        b1p = np.cumprod(np.full(n_steps, self.beta1, np.float32), dtype=np.float32)
        b2p = np.cumprod(np.full(n_steps, self.beta2, np.float32), dtype=np.float32)
        tab = np.zeros(n_steps + 1, np.float32)
        tab[1:] = (self.lr * np.sqrt(one - b2p) / (one - b1p)).astype(np.float32)
        return tab


def adam_sparse(var, m, v, grad, st):
    call("nrhip_adam_sparse_tf", _ptr(var, torch.float32), _ptr(m), _ptr(v), _ptr(grad),
         var.numel(), float(st.alpha()), float(st.beta1), float(st.beta2), float(st.eps),
         _stream())


def adam_dense(var, m, v, grad, st, clear_grad=True):
    call("nrhip_adam_dense_tf", _ptr(var, torch.float32), _ptr(m), _ptr(v), _ptr(grad),
         var.numel(), float(st.alpha()), float(st.beta1), float(st.beta2), float(st.eps),
         1 if clear_grad else 0, _stream())


def adam_dense_multi(tensors, st):
    """Dense TF-Adam on several (var, m, v, grad[, clear_grad]) tuples in one launch (<= 32 per launch)."""
    for lo in range(0, len(tensors), 32):
        part = tensors[lo:lo + 32]
        n = len(part)
        for t in part:
            for x in t[:4]:
                _ptr(x, torch.float32)                       # type / device / contiguity checks
        arr = lambda i: (C.c_void_p * n)(*[t[i].data_ptr() for t in part])
        sizes = (C.c_int64 * n)(*[t[0].numel() for t in part])
        clear = (C.c_int32 * n)(*[1 if (len(t) > 4 and t[4]) else 0 for t in part])
        call("nrhip_adam_dense_tf_multi", n, arr(0), arr(1), arr(2), arr(3), sizes, clear, st.alpha(),
             st.beta1, st.beta2, st.eps, _stream())


def adam_dense2(var, m, v, grad_a, grad_b, st):
    """Dense TF Adam with gradient grad_a + grad_b (both left untouched)."""
    call("nrhip_adam_dense_tf2", _ptr(var, torch.float32), _ptr(m), _ptr(v), _ptr(grad_a),
         _ptr(grad_b), var.numel(), float(st.alpha()), float(st.beta1), float(st.beta2),
         float(st.eps), _stream())


def rows_div(rows, src, denom, dst):
    call("nrhip_rows_div", _ptr(rows, torch.int32), rows.numel(), src.shape[1], _ptr(src),
         float(denom), _ptr(dst), _stream())


def rows_gather(rows, src, dst):
    """dst[w, :d] = src[rows[w], :d] (src and dst may be column blocks of wider row-major buffers)."""
    d = src.shape[1]
    call("nrhip_rows_gather_ld", _ptr(rows, torch.int32), rows.numel(), d, C.c_void_p(src.data_ptr()),
         src.stride(0), C.c_void_p(dst.data_ptr()), dst.stride(0), _stream())


def rows_gather2(rows, src_a, src_b, dst_a, dst_b):
    """dst_a[w] = src_a[rows[w]] and dst_b[w] = src_b[rows[w]] in one launch (column blocks of wider buffers allowed)"""
    d = src_a.shape[1]
    call("nrhip_rows_gather2", _ptr(rows, torch.int32), rows.numel(), d, C.c_void_p(src_a.data_ptr()), src_a.stride(0),
         C.c_void_p(src_b.data_ptr()), src_b.stride(0), C.c_void_p(dst_a.data_ptr()), dst_a.stride(0),
         C.c_void_p(dst_b.data_ptr()), dst_b.stride(0), _stream())


def partials_sum(parts, world, n, out):
    """out[k] = sum over ranks r ascending of parts[r][k] (nrhip_partials_sum)"""
    call("nrhip_partials_sum", _ptr(parts, torch.float32), int(world), int(n), _ptr(out, torch.float32), _stream())


def partials_sum_rows(parts, world, out=None, addend=None, sum_in=None, sum_out=None, row_mask=None):
    """out[r] = parts[0][r] + parts[1][r] + ... (rank order) (+ addend); sum_out = sum_in + out — the owner's half of
    the reduced-exchange hop (nrhip_partials_sum_rows); parts is [world][n_rows][d] contiguous."""
    n_rows, d = parts.shape[-2], parts.shape[-1]
    call("nrhip_partials_sum_rows", _ptr(parts, torch.float32), int(world), int(n_rows), int(d),
         _ptr(out, torch.float32, allow_none=True), _ptr(addend, allow_none=True), _ptr(sum_in, allow_none=True),
         _ptr(sum_out, allow_none=True), _ptr(row_mask, torch.uint8, allow_none=True), _stream())
    return out


def route_batch(users, pos, neg, n_users, bu, bi, code_base, world, keys, packed, order, inv, counts=None):
    """requests of a batch in owner order (nrhip_route_batch); all outputs are preallocated device tensors"""
    call("nrhip_route_batch", _ptr(users, torch.int32), _ptr(pos, torch.int32), _ptr(neg, torch.int32),
         users.numel(), int(n_users), int(bu), int(bi), int(code_base), _ptr(keys, torch.int64),
         _ptr(packed, torch.int32), _ptr(order, torch.int32), _ptr(inv, torch.int32),
         _ptr(counts, torch.int32, allow_none=True), int(world), _stream())


def route_owner_keys(rows, codes, recv_prefix, size_off, world, global_batch, code_base, keys, index_of_pos):
    """sorted (row, global position) keys of the rows this rank was asked for (nrhip_route_owner_keys)"""
    call("nrhip_route_owner_keys", _ptr(rows, torch.int32, allow_none=True), _ptr(codes, torch.int32, allow_none=True),
         keys.numel(), _ptr(recv_prefix, torch.int32), _ptr(size_off, torch.int32), int(world), int(global_batch),
         int(code_base), _ptr(keys, torch.int64), _ptr(index_of_pos, torch.int32), _stream())


def rows_scatter_add(rows, src, dst):
    """dst[rows[w]] += src[w, :d] (fp32 atomics; repeats summed)."""
    d = dst.shape[1]
    call("nrhip_rows_scatter_add", _ptr(rows, torch.int32), rows.numel(), d,
         C.c_void_p(src.data_ptr()), src.stride(0), _ptr(dst, torch.float32), _stream())


def rows_sum_sorted(keys, index_of_pos, src, dst):
    """dst[row] = ordered sum of src rows per run of the sorted int64 keys (row << 32 | position)."""
    call("nrhip_rows_sum_sorted", _ptr(keys, torch.int64), keys.numel(), _ptr(index_of_pos, torch.int32),
         dst.shape[1], C.c_void_p(src.data_ptr()), src.stride(0), _ptr(dst, torch.float32), _stream())


def rows_sum_sorted2(keys, index_of_pos, src_a, dst_a, src_b, dst_b):
    """two tables along the same runs in one launch (see rows_sum_sorted)"""
    call("nrhip_rows_sum_sorted2", _ptr(keys, torch.int64), keys.numel(), _ptr(index_of_pos, torch.int32),
         dst_a.shape[1], C.c_void_p(src_a.data_ptr()), src_a.stride(0), _ptr(dst_a, torch.float32),
         C.c_void_p(src_b.data_ptr()), src_b.stride(0), _ptr(dst_b, torch.float32), _stream())


def sort_keys(keys):
    """Ascending in-place sort of int64 keys (non-negative): one LDS bitonic workgroup up to 16384
    keys, the segmented multi-workgroup network beyond (csrc/bpr.hip)."""
    call("nrhip_sort_u64", _ptr(keys, torch.int64), keys.numel(), _stream())
    return keys


def rows_clear(rows, d, bufs=(), flag=None):
    b = list(bufs) + [None] * (4 - len(bufs))
    call("nrhip_rows_clear", _ptr(rows, torch.int32), rows.numel(), int(d),
         _ptr(b[0], allow_none=True), _ptr(b[1], allow_none=True), _ptr(b[2], allow_none=True),
         _ptr(b[3], allow_none=True), _ptr(flag, allow_none=True), _stream())


def bpr_plan(users, items, third, batch, n_users, out=None):
    """Batch plans of a whole stream of (user, item[, third]) ids cut into batches of `batch`:
    uint64 keys (row << 32 | position; item rows offset by n_users) sorted per batch — the order
    in which the gradient kernels add the occurrences of a row (TF's unsorted_segment_sum order).
    Returned as an int64 tensor of n_cls * len(users) keys; batch k's slice starts at
    n_cls * k * batch."""
    n = users.numel()
    n_cls = 2 if third is None else 3
    if out is None:
        out = torch.empty(max(n_cls * n, 1), dtype=torch.int64, device=users.device)
    call("nrhip_bpr_plan", _ptr(users, torch.int32), _ptr(items, torch.int32),
         _ptr(third, torch.int32, allow_none=True), n, int(batch), int(n_users),
         _ptr(out, torch.int64), _stream())
    return out[:n_cls * n]


def _work(terms, batch):
    if terms.numel() < 8 * batch:
        raise ValueError("work buffer holds %d floats, 8 * batch = %d needed" % (terms.numel(), 8 * batch))
    return _ptr(terms, torch.float32)


def bpr_mf_grad(P, Q, users, pos, neg, reg, GP, GQ, terms, loss2, plan=None):
    call("nrhip_bpr_mf_grad", _ptr(P, torch.float32), _ptr(Q, torch.float32), P.shape[1], P.shape[0],
         _ptr(users, torch.int32), _ptr(pos, torch.int32), _ptr(neg, torch.int32), users.numel(),
         float(reg), _ptr(GP), _ptr(GQ), _work(terms, users.numel()), _ptr(loss2),
         _ptr(plan, torch.int64, allow_none=True), _stream())


PAIRWISE_LOSSES = {"bpr": 0, "hinge": 1, "square": 2}                # learner.py:19-29
POINTWISE_LOSSES = {"cross_entropy": 0, "square": 1}                  # learner.py:31-41
ROW_OPTIMIZERS = {"gd": 0, "adagrad": 1, "rmsprop": 2, "momentum": 3}  # learner.py:2-16 (adam: adam_sparse)


def pairwise_mf_grad(P, Q, users, pos, neg, reg, loss, GP, GQ, terms, loss2, plan=None):
    call("nrhip_pairwise_mf_grad", _ptr(P, torch.float32), _ptr(Q, torch.float32), P.shape[1],
         P.shape[0], _ptr(users, torch.int32), _ptr(pos, torch.int32), _ptr(neg, torch.int32),
         users.numel(), float(reg), PAIRWISE_LOSSES[loss], _ptr(GP), _ptr(GQ),
         _work(terms, users.numel()), _ptr(loss2), _ptr(plan, torch.int64, allow_none=True), _stream())


def pointwise_mf_grad(P, Q, users, items, labels, reg, loss, GP, GQ, terms, loss2, plan=None):
    call("nrhip_pointwise_mf_grad", _ptr(P, torch.float32), _ptr(Q, torch.float32), P.shape[1],
         P.shape[0], _ptr(users, torch.int32), _ptr(items, torch.int32), _ptr(labels, torch.float32),
         users.numel(), float(reg), POINTWISE_LOSSES[loss], _ptr(GP), _ptr(GQ),
         _work(terms, users.numel()), _ptr(loss2), _ptr(plan, torch.int64, allow_none=True), _stream())


def gather_u8(src, index, dst):
    """dst[i] = src[index[i]] (uint8)"""
    call("nrhip_gather_u8", _ptr(src, torch.uint8), _ptr(index, torch.int32), index.numel(), _ptr(dst, torch.uint8),
         _stream())


def mark_rows(ids, flag, offset=0):
    call("nrhip_mark_rows", _ptr(ids, torch.int32), ids.numel(), int(offset), _ptr(flag, torch.uint8),
         _stream())


def optimizer_rows(kind, var, slot0, slot1, grad, flag, lr, hyper1=0.0, hyper2=0.0, eps=0.0):
    """TF-1.12 sparse update of the flagged rows (learner.py:2-16); clears grad rows and flags."""
    call("nrhip_optimizer_rows_tf", ROW_OPTIMIZERS[kind], _ptr(var, torch.float32),
         _ptr(slot0, torch.float32, allow_none=True), _ptr(slot1, torch.float32, allow_none=True),
         _ptr(grad, torch.float32), _ptr(flag, torch.uint8), var.shape[0], var.shape[1], float(lr),
         float(hyper1), float(hyper2), float(eps), _stream())


class DenseLearner:
    """util/learner.py:2-17 for DENSE variables (NGCF / Mult-VAE: every trainable feeds a dense op, so TF applies
    its Apply* kernels): gd, adagrad (initial_accumulator_value 1e-8, learner.py:5-6), rmsprop
    (tf.train.RMSPropOptimizer(lr): decay 0.9, momentum 0, epsilon 1e-10; `rms` starts at ones), momentum
    (learner.py:13-14, 0.9).  The engines hand in their Adam moment buffers as slots — `init_slots` gives them the
    optimiser's initial values; Adam itself stays on its own kernels."""

    KINDS = ("gd", "adagrad", "rmsprop", "momentum")

    def __init__(self, kind, lr, momentum=0.9):
        kind = str(kind).lower()
        if kind not in self.KINDS:
            raise ValueError("please select a suitable optimizer")            # learner.py:15-16
        self.kind, self.lr, self.momentum = kind, float(lr), float(momentum)

    def init_slots(self, slot0, slot1):
        """slot tensors (any iterable each) set to what TF creates them with"""
        v0 = {"gd": 0.0, "adagrad": 1e-8, "rmsprop": 1.0, "momentum": 0.0}[self.kind]
        for t in slot0:
            t.fill_(v0)
        for t in slot1:
            t.zero_()

    def apply(self, tensors):
        """tensors: (var, slot0, slot1, grad[, clear_grad]) tuples, as adam_dense_multi takes them"""
        h1, h2, eps = {"gd": (0.0, 0.0, 0.0), "adagrad": (0.0, 0.0, 0.0), "rmsprop": (0.9, 0.0, 1e-10),
                       "momentum": (self.momentum, 0.0, 0.0)}[self.kind]
        for t in tensors:
            var, s0, s1, grad = t[:4]
            call("nrhip_optimizer_dense_tf", ROW_OPTIMIZERS[self.kind], _ptr(var, torch.float32),
                 _ptr(s0, torch.float32, allow_none=True), _ptr(s1, torch.float32, allow_none=True),
                 _ptr(grad, torch.float32), var.numel(), self.lr, h1, h2, eps, 1 if (len(t) > 4 and t[4]) else 0,
                 _stream())


def make_learner(learner, lr):
    """None for adam (the engines' own kernels), a DenseLearner for the other learners of learner.py"""
    return None if str(learner).lower() == "adam" else DenseLearner(learner, lr)


def lightgcn_mark_batch(users, pos, neg, n_users, rows_out, row_flag):
    """rows_out[3B] = users | n_users+pos | n_users+neg; row_flag[those] = 1."""
    call("nrhip_lightgcn_mark_batch", _ptr(users, torch.int32), _ptr(pos, torch.int32),
         _ptr(neg, torch.int32), users.numel(), int(n_users), _ptr(rows_out, torch.int32),
         _ptr(row_flag, torch.uint8), _stream())


def lightgcn_bpr_grad(Esum, E0, n_users, n_layers, users, pos, neg, reg, Gstar, Greg, terms, loss2,
                      plan=None, divided=False):
    """divided=True (only when n_layers + 1 is a power of two): Gstar receives dLoss/dE* already divided by
    n_layers + 1 — exact, every term divided instead of the sum (nrhip_lightgcn_bpr_grad_h)"""
    call("nrhip_lightgcn_bpr_grad_h" if divided else "nrhip_lightgcn_bpr_grad", _ptr(Esum, torch.float32), _ptr(E0, torch.float32), n_users,
         E0.shape[1], n_layers, _ptr(users, torch.int32), _ptr(pos, torch.int32),
         _ptr(neg, torch.int32), users.numel(), float(reg), _ptr(Gstar), _ptr(Greg),
         _work(terms, users.numel()), _ptr(loss2, allow_none=True),
         _ptr(plan, torch.int64, allow_none=True), _stream())


def scale(x, a, out):
    call("nrhip_scale", _ptr(x), float(a), _ptr(out), x.numel(), _stream())


def add(x, y, out):
    call("nrhip_add", _ptr(x), _ptr(y), _ptr(out), x.numel(), _stream())


def div_scalar(x, denom, out):
    call("nrhip_div_scalar", _ptr(x), float(denom), _ptr(out), x.numel(), _stream())


class SpmmCSR:
    """A CSR matrix resident on the device plus its row-segment plan."""

    def __init__(self, indptr, indices, vals, n_cols=None, item_rows=0, item_nnz=0, split_row=0):
        dev = require_gpu()
        if isinstance(indices, torch.Tensor):          # device-built matrices (synth.device_*): stay there
            self.indptr = indptr.to(dev, torch.int64).contiguous()
            self.h_indptr = self.indptr.cpu().numpy()
            self.indices = indices.to(dev, torch.int32).contiguous()
            self.vals = vals.to(dev, torch.float32).contiguous()
            self._h_indices = None                     # fetched only if a lane-group schedule is built
        else:
            self.h_indptr = np.ascontiguousarray(indptr, dtype=np.int64)
            idx = np.ascontiguousarray(indices, dtype=np.int32)
            val = np.ascontiguousarray(vals, dtype=np.float32)
            self._h_indices = idx
            self.indices = torch.from_numpy(idx if len(idx) else np.zeros(1, np.int32)).to(dev)
            self.vals = torch.from_numpy(val if len(val) else np.zeros(1, np.float32)).to(dev)
            self.indptr = torch.from_numpy(self.h_indptr).to(dev)
        self.n_rows = len(self.h_indptr) - 1
        self.nnz = int(self.h_indptr[-1])
        self.n_cols = self.n_rows if n_cols is None else n_cols
        self.split_row = int(split_row)
        # False: never attach the lane-group schedule (the sharded engine's column slices of a d >= 128 table keep the
        # work-item kernel's 256-non-zero segments, i.e. the single-GPU engine's association at that d)
        self.lane_group = True
        self.blocked = None               # lane-group schedule of the last dim asked for
        self._blocked = {}                # d -> (plan handle | None, buffer), built on first use
        nbytes = C.c_size_t(0)
        call("nrhip_spmm_plan_bytes", self.n_rows, self.nnz, C.byref(nbytes))
        self.plan_buf = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=dev)
        self.plan = C.c_void_p(0)
        call("nrhip_spmm_plan_create", self.h_indptr.ctypes.data_as(C.c_void_p), self.n_rows,
             int(item_rows), int(item_nnz), int(split_row), _ptr(self.plan_buf), self.plan_buf.numel(), _stream(), C.byref(self.plan))
        nseg, nsplit = C.c_int64(0), C.c_int64(0)
        call("nrhip_spmm_plan_info", self.plan, C.byref(nseg), C.byref(nsplit))
        self.n_segments, self.n_split_rows = nseg.value, nsplit.value
        self._ws = {}

    @property
    def h_indices(self):
        if self._h_indices is None:
            self._h_indices = self.indices[:self.nnz].cpu().numpy()
        return self._h_indices

    @staticmethod
    def from_scipy(mat, item_rows=0, item_nnz=0, split_row=0, keep_order=False):
        """keep_order: keep each row's storage order (the kernels sum a row in that order) instead of
        sorting the columns ascending — graph.lightgcn_adjacency(..., tf_order=True)"""
        m = mat.tocsr().astype(np.float32)
        if not keep_order:
            m.sort_indices()
        return SpmmCSR(m.indptr, m.indices, m.data, m.shape[1], item_rows, item_nnz, split_row)

    def __del__(self):
        try:
            for plan, _buf in getattr(self, "_blocked", {}).values():
                if plan is not None and plan.value:
                    _lib.lib.nrhip_spmm_blocked_plan_destroy(plan)
            self._blocked = {}
            if getattr(self, "plan", None) is not None and self.plan.value:
                _lib.lib.nrhip_spmm_plan_destroy(self.plan)
                self.plan = C.c_void_p(0)
        except Exception:
            pass

    def ensure_schedule(self, d, force=False):
        """Build the persistent lane-group schedule (spmm_blocked.hip) once and attach it to the
        plan.  Used for d <= 64 (1.3x faster than the work-item kernel at 64, 3-4x faster than the
        narrow-row segment kernel at 16); at d = 128 the
        two measure equal (92.7 vs 96.7 us per gowalla pass) and the work-item kernel keeps rows of
        up to 256 non-zeros in strict order, so 128 / 256 are attached only with force=True.
        Matrices the schedule does not fit keep the work-item kernel.  NEUREC_SPMM_BLOCKED=0
        disables it (A/B measurements)."""
        if d not in (16, 32, 64, 128, 256) or not self.lane_group or \
                (d in (128, 256) and not force and d not in self._blocked):
            return False
        if d not in self._blocked:
            self._blocked[d] = (None, None)
            if os.environ.get("NEUREC_SPMM_BLOCKED", "1") != "0" and self.nnz > 0:
                nbytes = C.c_size_t(0)
                call("nrhip_spmm_blocked_plan_bytes", self.n_rows, self.nnz, int(d), C.byref(nbytes))
                buf = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=self.indices.device)
                plan = C.c_void_p(0)
                try:
                    call("nrhip_spmm_blocked_plan_create", self.h_indptr.ctypes.data_as(C.c_void_p),
                         self.h_indices.ctypes.data_as(C.c_void_p), self.n_rows, self.split_row, d,
                         0, 0, 0, 0, 0, 0, _ptr(buf), buf.numel(), _stream(), C.byref(plan))
                    call("nrhip_spmm_plan_attach_blocked", self.plan, plan, d)
                    # the plan's own copy of the (column, value) pairs, in its row order: filled once
                    call("nrhip_spmm_blocked_pack", plan, _ptr(self.indices), _ptr(self.vals), _stream())
                    self._blocked[d] = (plan, buf)
                except NotImplementedError:
                    pass                   # does not fit the schedule: work-item kernel stays
        self.blocked = self._blocked[d][0]
        return self.blocked is not None

    def values_changed(self):
        """`vals` was rewritten in place (NGCF's node dropout): the lane-group schedules hold their own copy of the
        (column, value) pairs in schedule order and must re-read it"""
        for plan, _buf in self._blocked.values():
            if plan is not None and plan.value:
                call("nrhip_spmm_blocked_pack", plan, _ptr(self.indices), _ptr(self.vals), _stream())

    def exact_row_nnz(self, d):
        """Rows with at most this many non-zeros are summed strictly in ascending column order."""
        return 64 if self.ensure_schedule(d) else 256

    def _workspace(self, d):
        if d not in self._ws:
            nbytes = C.c_size_t(0)
            call("nrhip_spmm_workspace_bytes", self.plan, d, C.byref(nbytes))
            self._ws[d] = torch.empty(max(nbytes.value, 256), dtype=torch.uint8,
                                      device=self.indices.device)
        return self._ws[d]

    def matmul(self, X, out=None, addend=None, sum_in=None, sum_out=None, x_row_nonzero=None,
               y_row_wanted=None):
        """out = A @ X (+ addend); sum_out = sum_in + out (each optional).
        x_row_nonzero: optional uint8 [n_cols]; 0 promises that row of X is all zero (skipped).
        y_row_wanted: optional uint8 [n_rows]; rows with 0 are not produced (left untouched)."""
        d = X.shape[1]
        self.ensure_schedule(d)
        ws = self._workspace(d)
        if x_row_nonzero is None and y_row_wanted is None:
            call("nrhip_spmm_csr", self.plan, _ptr(self.indptr), _ptr(self.indices),
                 _ptr(self.vals), _ptr(X, torch.float32), d,
                 _ptr(out, torch.float32, allow_none=True), _ptr(addend, allow_none=True),
                 _ptr(sum_in, allow_none=True), _ptr(sum_out, allow_none=True), _ptr(ws),
                 ws.numel(), _stream())
        else:
            call("nrhip_spmm_csr_masked", self.plan, _ptr(self.indptr), _ptr(self.indices),
                 _ptr(self.vals), _ptr(X, torch.float32),
                 _ptr(x_row_nonzero, torch.uint8, allow_none=True),
                 _ptr(y_row_wanted, torch.uint8, allow_none=True), d,
                 _ptr(out, torch.float32, allow_none=True), _ptr(addend, allow_none=True),
                 _ptr(sum_in, allow_none=True), _ptr(sum_out, allow_none=True), _ptr(ws),
                 ws.numel(), _stream())
        return out

    def matmul_adam(self, X, addend, grad_b, var, m, v, st, row_flag=None):
        """The last backward hop of a step with TF's ApplyAdam as its epilogue (nrhip_spmm_csr_adam): row r of
        A @ X + addend + grad_b is consumed as var's gradient and never stored; with row_flag, rows of addend /
        grad_b whose flag is 0 are promised zero (not read) and the consumed rows and flags are cleared.
        Returns False when this matrix has no d = 64 lane-group schedule (the caller then runs matmul + Adam)."""
        d = X.shape[1]
        if d != 64 or not self.ensure_schedule(64):
            return False
        call("nrhip_spmm_csr_adam", self.plan, _ptr(self.indices), _ptr(self.vals), _ptr(X, torch.float32), d,
             _ptr(addend, torch.float32), _ptr(grad_b, torch.float32), _ptr(var, torch.float32), _ptr(m), _ptr(v),
             float(st.alpha()), float(st.beta1), float(st.beta2), float(st.eps), 1 if row_flag is not None else 0,
             _ptr(row_flag, torch.uint8, allow_none=True), _stream())
        return True

    def matmul_rows(self, X, rows, out=None, addend=None, sum_in=None, sum_out=None):
        """Only the listed rows (int32 device tensor, repeats allowed) of A @ X (+ epilogue)."""
        d = X.shape[1]
        call("nrhip_spmm_csr_rows", _ptr(self.indptr), _ptr(self.indices), _ptr(self.vals),
             _ptr(X, torch.float32), d, _ptr(rows, torch.int32), rows.numel(),
             _ptr(out, torch.float32, allow_none=True), _ptr(addend, allow_none=True),
             _ptr(sum_in, allow_none=True), _ptr(sum_out, allow_none=True), _stream())
        return out

    def algorithmic_bytes(self, d):
        """CSR idx+val read once, X read once, Y written once."""
        return self.nnz * 8 + (self.n_rows + 1) * 4 + 2 * self.n_rows * d * 4

    def masked_bytes(self, d):
        """A hop whose operand or result lives on the batch rows only: the CSR arrays and
        one dense [N][d] stream."""
        return self.nnz * 8 + (self.n_rows + 1) * 4 + self.n_rows * d * 4

    def full_pass_kernel(self, d):
        """Name (as rocprofv3 prints it) of the kernel an unmasked matmul at width d launches."""
        if self.ensure_schedule(d):
            return "spmm_blocked_kernel<false, 16, 8, %d, false>" % d
        return "spmm_item_kernel<%d, ...>" % d


def add2d(x, y, out):
    """out = x + y on 2-D views with unit inner stride (column blocks allowed)."""
    call("nrhip_add2d", C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(y.data_ptr()), y.stride(0),
         C.c_void_p(out.data_ptr()), out.stride(0), x.shape[0], x.shape[1], _stream())


def copy2d(x, out):
    call("nrhip_copy2d", C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(out.data_ptr()),
         out.stride(0), x.shape[0], x.shape[1], _stream())


def ngcf_layer_fwd(ego, S, W, keep, mask, mask_given, seed, step, layer, ego_out, out_block):
    """W = (W_gc, b_gc, W_bi, b_bi); out_block: this layer's column block of the concat output."""
    call("nrhip_ngcf_layer_fwd", _ptr(ego, torch.float32), _ptr(S, torch.float32), _ptr(W[0]),
         _ptr(W[1]), _ptr(W[2]), _ptr(W[3]), ego.shape[0], ego.shape[1], float(keep),
         _ptr(mask, torch.uint8), 1 if mask_given else 0, C.c_uint64(seed & (2**64 - 1)),
         C.c_uint64(step), int(layer), _ptr(ego_out, torch.float32),
         C.c_void_p(out_block.data_ptr()), out_block.stride(0), _stream())


def ngcf_layer_bwd(ego, S, W, keep, mask, dout_block, dego_next, dS, dego_direct, dT1, dT2, dW, ws):
    call("nrhip_ngcf_layer_bwd", _ptr(ego, torch.float32), _ptr(S, torch.float32), _ptr(W[0]),
         _ptr(W[1]), _ptr(W[2]), _ptr(W[3]), ego.shape[0], ego.shape[1], float(keep),
         _ptr(mask, torch.uint8), C.c_void_p(dout_block.data_ptr()), dout_block.stride(0),
         _ptr(dego_next, allow_none=True), _ptr(dS), _ptr(dego_direct), _ptr(dT1), _ptr(dT2),
         _ptr(dW[0]), _ptr(dW[1]), _ptr(dW[2]), _ptr(dW[3]), _ptr(ws), ws.numel(), _stream())


def ngcf_workspace(n_rows, device):
    nbytes = C.c_size_t(0)
    call("nrhip_ngcf_workspace_bytes", int(n_rows), C.byref(nbytes))
    return torch.empty(nbytes.value, dtype=torch.uint8, device=device)


class NativeStep:
    """Context of the native step drivers (csrc/step.hip): a record of device pointers owned by
    the Python engine object, which must outlive it."""

    def __init__(self, kind, handle, owner):
        self.kind, self.handle, self.owner = kind, handle, owner

    @staticmethod
    def for_lightgcn(eng):
        b = _lib.LightGCNBuffers()
        eng.A.ensure_schedule(eng.d)
        eng.At.ensure_schedule(eng.d)
        ws = eng.A._workspace(eng.d)
        wst = eng.At._workspace(eng.d)
        ws = ws if ws.numel() >= wst.numel() else wst
        vals = dict(plan=eng.A.plan.value, plan_t=eng.At.plan.value, indptr=eng.A.indptr,
                    indices=eng.A.indices, vals=eng.A.vals, indptr_t=eng.At.indptr,
                    indices_t=eng.At.indices, vals_t=eng.At.vals, E0=eng.E0, m=eng.m, v=eng.v,
                    Ea=eng.Ea, Eb=eng.Eb, Esum=eng.Esum, Esum_rows=eng.Esum_rows, Gstar=eng.Gstar,
                    Greg=eng.Greg, H=eng.H, Ga=eng.Ga, Gb=eng.Gb, batch_rows=eng.batch_rows,
                    row_flag=eng.row_flag, terms=eng.terms, spmm_ws=ws)
        for k, t in vals.items():
            setattr(b, k, t if isinstance(t, int) else t.data_ptr())
        b.spmm_ws_bytes, b.n_users, b.n_nodes = ws.numel(), eng.n_users, eng.N
        b.d, b.n_layers, b.max_batch, b.reg = eng.d, eng.n_layers, eng.max_batch, eng.reg
        h = C.c_void_p(0)
        call("nrhip_lightgcn_ctx_create", C.byref(b), C.byref(h))
        ctx = NativeStep("lightgcn", h, eng)
        ctx._keep = (ws,)
        return ctx

    @staticmethod
    def for_ngcf(eng):
        b = _lib.NGCFBuffers()
        eng.A.ensure_schedule(eng.d)
        eng.At.ensure_schedule(eng.d)
        ws = eng.A._workspace(eng.d)
        wst = eng.At._workspace(eng.d)
        ws = ws if ws.numel() >= wst.numel() else wst
        b.plan, b.plan_t = eng.A.plan.value, eng.At.plan.value
        for k, t in dict(indptr=eng.A.indptr, indices=eng.A.indices, vals=eng.A.vals, indptr_t=eng.At.indptr,
                         indices_t=eng.At.indices, vals_t=eng.At.vals, spmm_ws=ws, E0=eng.E0, mE=eng.mE,
                         vE=eng.vE, gE0=eng.gE0, Out=eng.Out, dOut=eng.dOut, dS=eng.dS, dEd=eng.dEd, dT1=eng.dT1,
                         dT2=eng.dT2, terms=eng.terms, rows=eng.rows, flag=eng.flag, ws=eng.ws).items():
            setattr(b, k, t.data_ptr())
        b.spmm_ws_bytes, b.ws_bytes = ws.numel(), eng.ws.numel()
        b.dEgo[0], b.dEgo[1] = eng.dEgo[0].data_ptr(), eng.dEgo[1].data_ptr()
        for k in range(eng.L + 1):
            b.ego[k] = eng.ego[k].data_ptr()
        for k in range(eng.L):
            b.S[k], b.mask[k] = eng.S[k].data_ptr(), eng.mask[k].data_ptr()
            for j in range(4):
                b.W[k][j], b.gW[k][j] = eng.W[k][j].data_ptr(), eng.gW[k][j].data_ptr()
                b.mW[k][j], b.vW[k][j] = eng.mW[k][j].data_ptr(), eng.vW[k][j].data_ptr()
        b.n_users, b.n_nodes, b.d, b.n_layers = eng.n_users, eng.N, eng.d, eng.L
        b.max_batch, b.reg, b.keep = eng.max_batch, eng.reg, eng.keep
        h = C.c_void_p(0)
        call("nrhip_ngcf_ctx_create", C.byref(b), C.byref(h))
        ctx = NativeStep("ngcf", h, eng)
        ctx._keep = (ws,)
        return ctx

    def ngcf_forward(self, seed, step_counter, mask_given):
        call("nrhip_ngcf_forward", self.handle, C.c_uint64(seed & (2**64 - 1)), C.c_uint64(step_counter),
             1 if mask_given else 0, _stream())

    def ngcf_step(self, users, pos, neg, st, seed, step_counter, mask_given, loss2, plan=None):
        call("nrhip_ngcf_step", self.handle, self._idx(users), self._idx(pos), self._idx(neg), users.numel(),
             self._plan(plan, 3 * users.numel()), C.c_uint64(seed & (2**64 - 1)), C.c_uint64(step_counter),
             1 if mask_given else 0, float(st.alpha()), float(st.beta1), float(st.beta2), float(st.eps),
             _ptr(loss2, torch.float32), _stream())

    @staticmethod
    def for_mf(eng):
        b = _lib.MFBuffers()
        for k in ("_P", "_Q", "terms"):
            setattr(b, k.lstrip("_"), getattr(eng, k).data_ptr())
        for k in ("mP", "vP", "mQ", "vQ"):
            setattr(b, k, eng._views[k].data_ptr())        # the raw views: the properties would flush
        if not getattr(eng, "fused", False):
            b.GP, b.GQ = eng.GP.data_ptr(), eng.GQ.data_ptr()
        b.n_users, b.n_items, b.d = eng._P.shape[0], eng._Q.shape[0], eng._P.shape[1]
        b.max_batch, b.reg = eng.max_batch, eng.reg
        if eng.lazy:
            b.alpha_tab = eng._alpha_tab.data_ptr()
            b.alpha_len, b.lazy_period = eng._alpha_tab.numel(), eng.lazy_period
            if eng.fused:
                b.tw, b.inb = eng._tw.data_ptr(), eng._inb.data_ptr()
            else:
                b.last, b.stamp = eng._last.data_ptr(), eng._stamp.data_ptr()
        h = C.c_void_p(0)
        call("nrhip_mf_ctx_create", C.byref(b), C.byref(h))
        return NativeStep("mf", h, eng)

    def __del__(self):
        try:
            if self.handle and self.handle.value:
                name = {"lightgcn": "nrhip_lightgcn_ctx_destroy", "ngcf": "nrhip_ngcf_ctx_destroy"}.get(
                    self.kind, "nrhip_mf_ctx_destroy")
                getattr(_lib.lib, name)(self.handle)
                self.handle = C.c_void_p(0)
        except Exception:
            pass

    @staticmethod
    def _idx(t):
        if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError("batch ids must be contiguous int32 device tensors")
        return C.c_void_p(t.data_ptr())

    @staticmethod
    def _plan(plan, n_occ):
        if plan is None:
            return C.c_void_p(0)
        if plan.dtype != torch.int64 or not plan.is_cuda or plan.numel() != n_occ:
            raise TypeError("batch plan must be the int64 device tensor of %d keys bpr_plan made" % n_occ)
        return C.c_void_p(plan.data_ptr())

    def lightgcn_step(self, users, pos, neg, st, loss2=None, plan=None):
        call("nrhip_lightgcn_step", self.handle, self._idx(users), self._idx(pos), self._idx(neg),
             users.numel(), self._plan(plan, 3 * users.numel()), float(st.alpha()), float(st.beta1),
             float(st.beta2), float(st.eps), _ptr(loss2, allow_none=True), _stream())

    def lightgcn_step_colshard_fwd(self, users, pos, neg, partials):
        call("nrhip_lightgcn_step_colshard_fwd", self.handle, self._idx(users), self._idx(pos), self._idx(neg),
             users.numel(), _ptr(partials, torch.float32), _stream())

    def lightgcn_step_colshard_bwd(self, users, pos, neg, st, loss2, plan, given):
        call("nrhip_lightgcn_step_colshard_bwd", self.handle, self._idx(users), self._idx(pos), self._idx(neg),
             users.numel(), self._plan(plan, 3 * users.numel()), _ptr(given, torch.float32), float(st.alpha()),
             float(st.beta1), float(st.beta2), float(st.eps), _ptr(loss2, torch.float32, allow_none=True), _stream())

    def lightgcn_step_grad(self, users, pos, neg, loss2, grad_out, plan=None):
        call("nrhip_lightgcn_step_grad", self.handle, self._idx(users), self._idx(pos),
             self._idx(neg), users.numel(), self._plan(plan, 3 * users.numel()),
             _ptr(loss2, allow_none=True), _ptr(grad_out, torch.float32), _stream())

    def lightgcn_step_apply(self, grad, st):
        call("nrhip_lightgcn_step_apply", self.handle, _ptr(grad, torch.float32),
             float(st.alpha()), float(st.beta1), float(st.beta2), float(st.eps), _stream())

    def mf_step(self, users, pos, neg, st, loss2, plan=None, next_plan=None):
        n_next = 0 if next_plan is None else next_plan.numel() // 3
        call("nrhip_mf_step", self.handle, self._idx(users), self._idx(pos), self._idx(neg),
             users.numel(), self._plan(plan, 3 * users.numel()), self._plan(next_plan, 3 * n_next), n_next,
             st.t + 1, float(st.alpha()), float(st.beta1), float(st.beta2), float(st.eps),
             _ptr(loss2, torch.float32), _stream())

    def mf_steps(self, users, pos, neg, batch, st, h_alpha, loss_steps, plans=None, terms_steps=None):
        """the consecutive batches of an epoch stream in one native call (nrhip_mf_steps); terms_steps:
        float32 device tensor of 2 * batch floats per step (one loss reduction launch for the whole call)"""
        n = users.numel()
        n_steps = (n + batch - 1) // batch
        if h_alpha.dtype != np.float32 or h_alpha.size < n_steps or not h_alpha.flags["C_CONTIGUOUS"]:
            raise TypeError("h_alpha: contiguous float32 host array of %d step sizes" % n_steps)
        if loss_steps.numel() < 2 * n_steps:
            raise ValueError("loss buffer holds %d floats, 2 per step = %d needed" % (loss_steps.numel(), 2 * n_steps))
        call("nrhip_mf_steps", self.handle, self._idx(users), self._idx(pos), self._idx(neg), n, int(batch),
             self._plan(plans, 3 * n), st.t + 1, h_alpha.ctypes.data_as(C.c_void_p), float(st.beta1),
             float(st.beta2), float(st.eps), _ptr(loss_steps, torch.float32),
             _ptr(terms_steps, torch.float32, allow_none=True), _stream())

    def mf_flush(self, st):
        call("nrhip_mf_flush", self.handle, st.t, float(st.beta1), float(st.beta2), float(st.eps),
             _stream())


def device_info():
    cu, clk, mem = C.c_int(0), C.c_int(0), C.c_size_t(0)
    name = C.create_string_buffer(128)
    call("nrhip_device_info", C.byref(cu), C.byref(clk), C.byref(mem), name, 128)
    return {"cu_count": cu.value, "clock_khz": clk.value, "hbm_bytes": mem.value,
            "name": name.value.decode()}


# ----------------------------------------------------------------------------- Mult-VAE
VAE_ACTS = {"tanh": 0, "sigmoid": 1, "relu": 2, "identity": 3}


def vae_encode(csr, rows, Wq0, bq0, Wq1, bq1, Wp0, bp0, act, keep, is_training, seed, step, bufs,
               drop_given=None, eps_given=None, h0val=None):
    """bufs = (H1, MU, LOGVAR, EPSSTD, ZS, G1, KLb) — see nrhip_vae_encode."""
    h, z = Wq0.shape[1], Wp0.shape[0]
    call("nrhip_vae_encode", _ptr(csr.indptr), _ptr(csr.indices), _ptr(rows, torch.int32),
         rows.numel(), h, z, _ptr(Wq0, torch.float32), _ptr(bq0), _ptr(Wq1), _ptr(bq1), _ptr(Wp0),
         _ptr(bp0), VAE_ACTS[act], float(keep), _ptr(drop_given, torch.float32, allow_none=True),
         _ptr(eps_given, torch.float32, allow_none=True), float(is_training),
         C.c_uint64(seed & (2**64 - 1)), C.c_uint64(step), _ptr(h0val, torch.float32, allow_none=True),
         *[_ptr(b, torch.float32) for b in bufs], _stream())


def vae_step_native(eng, rows, anneal, keep, drop_given, eps_given, want_loss, apply):
    """nrhip_vae_step: one native call for the narrow Mult-VAE step of `eng` (trainer.MultiVAEEngine); the argument
    block is filled once per engine (its buffers never move), the per-call values travel as arguments."""
    from ._lib import VaeStepArgs
    a = eng._step_args
    # the block holds raw device pointers: keyed on the data_ptr() of EVERY tensor in it (id() misses a
    # storage swapped in place and can be reused by a new tensor) — a moved or replaced buffer refills the block
    bufs = ("H1", "MU", "LOGVAR", "EPSSTD", "ZS", "G1", "KLb", "h0val", "nll", "dG1", "DA3", "DH2", "DA1", "stats",
            "regsum", "ws")
    key = tuple(t.data_ptr() for d in (eng.P, eng.G, eng.M, eng.V) for t in d.values()) + \
        tuple(getattr(eng, f).data_ptr() for f in bufs) + (eng.csr.indptr.data_ptr(), eng.csr.indices.data_ptr())
    if a is None or eng._step_key != key:
        eng._step_key = key
        a = VaeStepArgs()
        names = eng.NAMES
        for t in list(eng.P.values()) + list(eng.G.values()) + list(eng.M.values()) + list(eng.V.values()):
            _ptr(t, torch.float32)
        a.indptr, a.indices = eng.csr.indptr.data_ptr(), eng.csr.indices.data_ptr()
        a.n_items, a.h, a.z, a.act = eng.n_items, eng.h, eng.z, VAE_ACTS[eng.act]
        for k, n in enumerate(names):
            a.P[k], a.G[k], a.M[k], a.V[k] = (d[n].data_ptr() for d in (eng.P, eng.G, eng.M, eng.V))
            a.sizes[k] = eng.P[n].numel()
        for f in bufs:
            setattr(a, f, getattr(eng, f).data_ptr())
        a.ws_bytes = eng.ws.numel()
        a.reg, a.beta1, a.beta2, a.adam_eps, a.seed = eng.reg, eng.adam.beta1, eng.adam.beta2, eng.adam.eps, eng.seed & (2**64 - 1)
        eng._step_args = a
    B = rows.numel()
    for g in (drop_given, eps_given):
        if g is not None and (g.dtype != torch.float32 or not g.is_contiguous() or not g.is_cuda):
            raise ValueError("given dropout / noise tensors: contiguous float32 device tensors")
    a.drop_given = drop_given.data_ptr() if drop_given is not None else None
    a.eps_given = eps_given.data_ptr() if eps_given is not None else None
    call("nrhip_vae_step", C.byref(a), _ptr(rows, torch.int32), B, float(anneal), float(keep), float(eng.adam.alpha()),
         C.c_uint64(eng.t), 1 if want_loss else 0, 1 if apply else 0, _stream())


def add_row_bias(S, cols, bias):
    call("nrhip_add_row_bias", _ptr(S, torch.float32), S.stride(0), S.shape[0], cols, _ptr(bias),
         _stream())


def vae_workspace(batch, cols, device):
    nbytes = C.c_size_t(0)
    call("nrhip_vae_workspace_bytes", batch, cols, C.byref(nbytes))
    return torch.empty(nbytes.value, dtype=torch.uint8, device=device)


def vae_decoder_loss_grad(S, cols, bp1, csr, rows, G1, Wp1, nll, dWp1, dbp1, dG1, ws):
    call("nrhip_vae_decoder_loss_grad", _ptr(S, torch.float32), S.stride(0), rows.numel(), cols,
         G1.shape[1], _ptr(bp1), _ptr(csr.indptr), _ptr(csr.indices), _ptr(rows, torch.int32),
         _ptr(G1), _ptr(Wp1), _ptr(nll), _ptr(dWp1), _ptr(dbp1), _ptr(dG1), _ptr(ws), ws.numel(),
         _stream())


def vae_fused_workspace(batch, cols, device):
    nbytes = C.c_size_t(0)
    call("nrhip_vae_decoder_fused_workspace_bytes", batch, cols, C.byref(nbytes))
    return torch.empty(nbytes.value, dtype=torch.uint8, device=device)


def vae_decoder_fused(cols, bp1, csr, rows, G1, Wp1, nll, dWp1, dbp1, dG1, ws, dbg_logits=None):
    """nrhip_vae_decoder_fused: loss + gradients of the Mult-VAE decoder, logits never stored"""
    call("nrhip_vae_decoder_fused", rows.numel(), cols, G1.shape[1], _ptr(G1, torch.float32), _ptr(Wp1, torch.float32),
         _ptr(bp1), _ptr(csr.indptr), _ptr(csr.indices), _ptr(rows, torch.int32), _ptr(nll), _ptr(dWp1), _ptr(dbp1),
         _ptr(dG1), _ptr(ws), ws.numel(), _ptr(dbg_logits, torch.float32, allow_none=True), _stream())


def vae_mid_backward(batch, act, anneal, dG1, G1, H1, MU, LOGVAR, EPSSTD, ZS, Wp0, Wq1, DA3, DH2,
                     DA1, dWp0, dbp0, dWq1, dbq1, dbq0):
    z, h = Wp0.shape
    call("nrhip_vae_mid_backward", batch, h, z, VAE_ACTS[act], float(anneal),
         *[_ptr(t, torch.float32) for t in (dG1, G1, H1, MU, LOGVAR, EPSSTD, ZS, Wp0, Wq1, DA3, DH2,
                                            DA1, dWp0, dbp0, dWq1, dbq1, dbq0)], _stream())


def vae_dwq0(csr, rows, h0val, DA1, dWq0):
    call("nrhip_vae_dwq0", _ptr(csr.indptr), _ptr(csr.indices), _ptr(rows, torch.int32),
         rows.numel(), DA1.shape[1], _ptr(h0val, torch.float32), _ptr(DA1), _ptr(dWq0), _stream())


def axpy(a, x, y):
    call("nrhip_axpy", float(a), _ptr(x, torch.float32), _ptr(y, torch.float32), x.numel(), _stream())


def sumsq_accumulate(x, out_f64):
    call("nrhip_sumsq_accumulate", _ptr(x, torch.float32), x.numel(), _ptr(out_f64, torch.float64),
         _stream())


def mean2_f32(x, y, out):
    """out[0] = mean(x), out[1] = mean(y) — one launch"""
    call("nrhip_mean2_f32", _ptr(x, torch.float32), _ptr(y, torch.float32), x.numel(), _ptr(out, torch.float32), _stream())


def mean_f32(x, out):
    call("nrhip_mean_f32", _ptr(x, torch.float32), x.numel(), _ptr(out, torch.float32), _stream())
