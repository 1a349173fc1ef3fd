"""ctypes binding of libneurec_hip.so (the C ABI declared in include/neurec_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load,
importing this module raises.  Build it with ``python -m neurec_amd.build``.
"""
import ctypes as C
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libneurec_hip.so")

ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_WORKSPACE = 1, 2, 3, 4


class NeuRecHipError(RuntimeError):
    """A HIP runtime failure or workspace error reported by the native library."""


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "libneurec_hip.so is not built (expected at %s). Run `python -m neurec_amd.build` "
            "(needs hipcc; the engine has no CPU fallback)." % LIB_PATH)
    if os.path.isfile(_build.STAMP) and not _build.is_current() and not os.environ.get("NEUREC_ALLOW_STALE_LIB"):
        raise ImportError(
            "libneurec_hip.so was built from other sources than neurec_amd/csrc now holds (digest in %s differs). "
            "Run `python -m neurec_amd.build`; NEUREC_ALLOW_STALE_LIB=1 loads it anyway." % _build.STAMP)
    try:
        return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover - depends on the box
        raise ImportError("cannot load %s: %s" % (LIB_PATH, e))


lib = _load()

p = C.c_void_p
i32, i64, u64, f32 = C.c_int, C.c_int64, C.c_uint64, C.c_float
sz = C.c_size_t
psz = C.POINTER(C.c_size_t)



class LightGCNBuffers(C.Structure):
    """nrhip_lightgcn_buffers (include/neurec_hip.h)"""
    _fields_ = [(n, C.c_void_p) for n in (
        "plan", "plan_t", "indptr", "indices", "vals", "indptr_t", "indices_t", "vals_t",
        "E0", "m", "v", "Ea", "Eb", "Esum", "Esum_rows", "Gstar", "Greg", "H", "Ga", "Gb",
        "batch_rows", "row_flag", "terms", "spmm_ws")] + [
        ("spmm_ws_bytes", C.c_size_t), ("n_users", C.c_int), ("n_nodes", C.c_int), ("d", C.c_int),
        ("n_layers", C.c_int), ("max_batch", C.c_int), ("reg", C.c_float)]


class MFBuffers(C.Structure):
    """nrhip_mf_buffers (include/neurec_hip.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("P", "Q", "mP", "vP", "mQ", "vQ", "GP", "GQ", "terms")] + [
        ("n_users", C.c_int), ("n_items", C.c_int), ("d", C.c_int), ("max_batch", C.c_int),
        ("reg", C.c_float), ("last", C.c_void_p), ("stamp", C.c_void_p), ("alpha_tab", C.c_void_p),
        ("alpha_len", C.c_int),
        ("lazy_period", C.c_int), ("tw", C.c_void_p), ("inb", C.c_void_p)]


NGCF_MAX_LAYERS = 4


class NGCFBuffers(C.Structure):
    """nrhip_ngcf_buffers (include/neurec_hip.h)"""
    _L = NGCF_MAX_LAYERS
    _fields_ = [(n, C.c_void_p) for n in ("plan", "plan_t", "indptr", "indices", "vals", "indptr_t", "indices_t",
                                          "vals_t", "spmm_ws")] + [("spmm_ws_bytes", C.c_size_t)] + \
        [(n, C.c_void_p) for n in ("E0", "mE", "vE", "gE0", "Out", "dOut")] + \
        [("S", C.c_void_p * _L), ("ego", C.c_void_p * (_L + 1)), ("mask", C.c_void_p * _L),
         ("W", (C.c_void_p * 4) * _L), ("gW", (C.c_void_p * 4) * _L), ("mW", (C.c_void_p * 4) * _L),
         ("vW", (C.c_void_p * 4) * _L)] + \
        [(n, C.c_void_p) for n in ("dS", "dEd", "dT1", "dT2")] + [("dEgo", C.c_void_p * 2)] + \
        [("terms", C.c_void_p), ("rows", C.c_void_p), ("flag", C.c_void_p), ("ws", C.c_void_p),
         ("ws_bytes", C.c_size_t), ("n_users", C.c_int), ("n_nodes", C.c_int), ("d", C.c_int),
         ("n_layers", C.c_int), ("max_batch", C.c_int), ("reg", C.c_float), ("keep", C.c_float)]


NGCF_WIDE_MAX_LAYERS = 8


class NGCFWideBuffers(C.Structure):
    """nrhip_ngcf_wide_buffers (include/neurec_hip.h)"""
    _L = NGCF_WIDE_MAX_LAYERS
    _fields_ = [(n, C.c_void_p) for n in ("plan", "indptr", "indices", "vals", "plan_t", "indptr_t", "indices_t",
                                          "vals_t")] + \
        [("ws_fwd", C.c_void_p * _L), ("ws_fwd_bytes", C.c_size_t * _L), ("ws_bwd", C.c_void_p * _L),
         ("ws_bwd_bytes", C.c_size_t * _L)] + \
        [(n, C.c_int) for n in ("n_users", "n_nodes", "n_layers", "max_batch", "dsum", "splits")] + \
        [("w", C.c_int * (_L + 1)), ("wp", C.c_int * (_L + 1)), ("off", C.c_int * (_L + 2))] + \
        [(n, C.c_void_p) for n in ("E0p", "mE", "vE", "gE0", "Out", "dOut")] + \
        [("ego", C.c_void_p * (_L + 1)), ("S", C.c_void_p * _L), ("X2", C.c_void_p * _L), ("T1", C.c_void_p * _L),
         ("T2", C.c_void_p * _L), ("mask", C.c_void_p * _L),
         ("W", (C.c_void_p * 4) * _L), ("gW", (C.c_void_p * 4) * _L), ("mW", (C.c_void_p * 4) * _L),
         ("vW", (C.c_void_p * 4) * _L),
         ("dS", C.c_void_p * _L), ("dEd", C.c_void_p * _L), ("dEgo", C.c_void_p * _L)] + \
        [(n, C.c_void_p) for n in ("dT1", "dT2", "Y1", "Y2", "terms", "cs_ws")] + \
        [("cs_ws_bytes", C.c_size_t), ("rows", C.c_void_p), ("flag", C.c_void_p), ("gemm_ws", C.c_void_p),
         ("gemm_ws_bytes", C.c_size_t), ("reg", C.c_float), ("keep", C.c_float)]


class VaeStepArgs(C.Structure):
    """nrhip_vae_step_args (include/neurec_hip.h)"""
    _fields_ = [("indptr", C.c_void_p), ("indices", C.c_void_p), ("n_items", C.c_int), ("h", C.c_int), ("z", C.c_int),
                ("act", C.c_int), ("P", C.c_void_p * 8), ("G", C.c_void_p * 8), ("M", C.c_void_p * 8),
                ("V", C.c_void_p * 8), ("sizes", C.c_int64 * 8)] + \
        [(n, C.c_void_p) for n in ("H1", "MU", "LOGVAR", "EPSSTD", "ZS", "G1", "KLb", "h0val", "nll", "dG1", "DA3",
                                   "DH2", "DA1", "stats", "regsum", "ws")] + \
        [("ws_bytes", C.c_size_t), ("drop_given", C.c_void_p), ("eps_given", C.c_void_p), ("reg", C.c_float),
         ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float), ("seed", C.c_uint64)]


class EvalPrunedArgs(C.Structure):
    """nrhip_eval_pruned_args (include/neurec_hip.h)"""
    _fields_ = [("P", C.c_void_p), ("ldp", C.c_int64), ("Q", C.c_void_p), ("ldq", C.c_int64), ("d", C.c_int),
                ("cols", C.c_int), ("users", C.c_void_p), ("n_users", C.c_int), ("batch_rows", C.c_int),
                ("tr_indptr", C.c_void_p), ("tr_indices", C.c_void_p), ("truth_indptr", C.c_void_p),
                ("truth_indices", C.c_void_p), ("chunk_tile", C.c_void_p), ("chunk_begin", C.c_void_p),
                ("n_chunks", C.c_int), ("tile_ptr", C.c_void_p), ("plan_user", C.c_void_p), ("plan_mask", C.c_void_p),
                ("row_of", C.c_void_p), ("metric_ids", C.POINTER(C.c_int)), ("n_metric", C.c_int), ("top_k", C.c_int),
                ("n_keep", C.c_int), ("use_filter", C.c_int), ("prepare_items", C.c_int),
                ("gemm_ws", C.c_void_p), ("gemm_ws_bytes", C.c_size_t), ("filter_ws", C.c_void_p),
                ("filter_ws_bytes", C.c_size_t), ("tiles_ws", C.c_void_p), ("tiles_ws_bytes", C.c_size_t),
                ("M", C.c_void_p), ("mld", C.c_int64), ("eps", C.c_void_p), ("out", C.c_void_p), ("flags", C.c_void_p),
                ("sums", C.c_void_p), ("colsum_ws", C.c_void_p), ("colsum_ws_bytes", C.c_size_t)]


class EvalRedoArgs(C.Structure):
    """nrhip_eval_redo_args (include/neurec_hip.h)"""
    _fields_ = [("ev", C.POINTER(EvalPrunedArgs)), ("n_flagged", C.c_int), ("reload_items", C.c_int),
                ("scores", C.c_void_p), ("lds", C.c_int64), ("slab_rows", C.c_int), ("rows", C.c_void_p),
                ("row_users", C.c_void_p), ("count", C.c_void_p), ("fixed", C.c_void_p), ("ws", C.c_void_p),
                ("ws_bytes", C.c_size_t)]


# name -> argtypes; every function returns int status except where noted.
SIGNATURES = {
    "nrhip_device_info": [C.POINTER(i32), C.POINTER(i32), psz, C.c_char_p, i32],
    "nrhip_eval_workspace_bytes": [i32, i32, psz],
    "nrhip_mask_train": [p, i64, p, i32, i32, p, p, p],
    "nrhip_eval_scores": [p, i64, i32, i32, p, p, p, C.POINTER(i32), i32, i32, p, p, p, p, sz, p],
    "nrhip_eval_any_k_workspace_bytes": [i32, i32, i32, psz],
    "nrhip_eval_scores_any_k": [p, i64, i32, i32, p, p, p, C.POINTER(i32), i32, i32, p, p, p, sz, p],
    "nrhip_arg_topk": [p, i64, i32, i32, i32, p, p, p, sz, p],
    "nrhip_colsum_workspace_bytes": [i32, i32, psz],
    "nrhip_colsum_f64": [p, i64, i32, i32, p, p, sz, p],
    "nrhip_score_gemm_workspace_bytes": [i32, i32, i32, psz],
    "nrhip_score_gemm_prepare_items": [p, i64, i32, i32, p, sz, p],
    "nrhip_score_tilemax": [p, i64, p, i32, i32, i32, p, p, p, i64, p, sz, p],
    "nrhip_tile_strike_plan": [p, p, i32, i32, p, p, p, p, p, p, p, sz, p],
    "nrhip_score_tilemax_fix": [p, i64, i32, i32, p, p, i32, p, p, p, p, i32, i32, p, i64, p, sz, p],
    "nrhip_score_filter_workspace_bytes": [i32, i32, i32, psz],
    "nrhip_score_filter_kappa": [i32, p],
    "nrhip_score_filter_prepare_items": [p, i64, i32, i32, p, sz, i32, p],
    "nrhip_score_filter_tilemax": [p, i64, p, i32, i32, i32, p, i64, p, p, sz, i32, p],
    "nrhip_score_filter_i8_workspace_bytes": [i32, i32, i32, psz],
    "nrhip_score_filter_i8_prepare_items": [p, i64, i32, i32, p, sz, i32, p],
    "nrhip_score_filter_i8_tilemax": [p, i64, p, i32, i32, i32, p, i64, p, p, sz, i32, p],
    "nrhip_eval_tiles_workspace_bytes": [i32, i32, psz],
    "nrhip_eval_tiles": [p, i64, p, i64, p, i32, p, i32, i32, p, p, p, p, p, i32, i32, p, p, p, sz, p],
    "nrhip_eval_tiles_bounded_workspace_bytes": [i32, i32, i32, i32, psz],
    "nrhip_eval_tiles_bounded": [p, i64, p, i32, p, i64, p, i32, p, i32, i32, p, p, p, p, p, i32, i32, p, p, p, sz, p],
    "nrhip_eval_pruned": [C.POINTER(EvalPrunedArgs), p],
    "nrhip_eval_redo": [C.POINTER(EvalRedoArgs), p],
    "nrhip_vae_step": [C.POINTER(VaeStepArgs), p, i32, f32, f32, f32, u64, i32, i32, p],
    "nrhip_score_gemm_items_kmajor": [p, i32, i32, p, p],
    "nrhip_score_gemm": [p, i64, p, i32, i32, i32, p, i64, p, sz, p],
    "nrhip_sample_bpr_epoch": [p, p, p, i64, i32, i32, u64, u64, i32, i64, i64, p, p, p, p],
    "nrhip_sample_instances_epoch": [p, p, p, p, p, p, p, i64, i32, i32, i32, i32, u64, u64, i32, i64, i64, p, p, p, p, p, p],
    "nrhip_randint_choice_batch": [i32, i32, i64, p, p, p, i32, u64, u64, p, p],
    "nrhip_bpr_plan": [p, p, p, i64, i32, i32, p, p],
    "nrhip_bpr_mf_grad": [p, p, i32, i32, p, p, p, i32, f32, p, p, p, p, p, p],
    "nrhip_adam_sparse_tf": [p, p, p, p, i64, f32, f32, f32, f32, p],
    "nrhip_adam_dense_tf": [p, p, p, p, i64, f32, f32, f32, f32, i32, p],
    "nrhip_adam_dense_tf2": [p, p, p, p, p, i64, f32, f32, f32, f32, p],
    "nrhip_adam_dense_tf_multi": [i32, p, p, p, p, p, p, f32, f32, f32, f32, p],
    "nrhip_rows_div": [p, i32, i32, p, f32, p, p],
    "nrhip_rows_clear": [p, i32, i32, p, p, p, p, p, p],
    "nrhip_spmm_plan_bytes": [i64, i64, psz],
    "nrhip_spmm_plan_create": [p, i64, i32, i32, i64, p, sz, p, C.POINTER(p)],
    "nrhip_spmm_plan_destroy": [p],
    "nrhip_spmm_plan_info": [p, C.POINTER(i64), C.POINTER(i64)],
    "nrhip_spmm_workspace_bytes": [p, i32, psz],
    "nrhip_spmm_csr": [p, p, p, p, p, i32, p, p, p, p, p, sz, p],
    "nrhip_spmm_csr_masked": [p, p, p, p, p, p, p, i32, p, p, p, p, p, sz, p],
    "nrhip_spmm_csr_carry": [p, p, p, p, p, p, i32, i32, p, i32, p, p],
    "nrhip_spmm_chunks_finish": [p, i64, p, i32, p, p, p, p, p, p],
    "nrhip_partials_sum_rows": [p, i32, i64, i32, p, p, p, p, p, p],
    "nrhip_spmm_csr_rows": [p, p, p, p, i32, p, i32, p, p, p, p, p],
    "nrhip_lightgcn_mark_batch": [p, p, p, i32, i32, p, p, p],
    "nrhip_lightgcn_bpr_grad": [p, p, i32, i32, i32, p, p, p, i32, f32, p, p, p, p, p, p],
    "nrhip_lightgcn_bpr_grad_h": [p, p, i32, i32, i32, p, p, p, i32, f32, p, p, p, p, p, p],
    "nrhip_lightgcn_ctx_create": [C.POINTER(LightGCNBuffers), C.POINTER(p)],
    "nrhip_lightgcn_ctx_destroy": [p],
    "nrhip_lightgcn_step": [p, p, p, p, i32, p, f32, f32, f32, f32, p, p],
    "nrhip_lightgcn_step_grad": [p, p, p, p, i32, p, p, p, p],
    "nrhip_lightgcn_step_apply": [p, p, f32, f32, f32, f32, p],
    "nrhip_mf_ctx_create": [C.POINTER(MFBuffers), C.POINTER(p)],
    "nrhip_mf_ctx_destroy": [p],
    "nrhip_mf_step": [p, p, p, p, i32, p, p, i32, i32, f32, f32, f32, f32, p, p],
    "nrhip_mf_flush": [p, i32, f32, f32, f32, p],
    "nrhip_ngcf_ctx_create": [C.POINTER(NGCFBuffers), C.POINTER(p)],
    "nrhip_ngcf_ctx_destroy": [p],
    "nrhip_ngcf_forward": [p, C.c_uint64, C.c_uint64, i32, p],
    "nrhip_ngcf_step": [p, p, p, p, i32, p, C.c_uint64, C.c_uint64, i32, f32, f32, f32, f32, p, p],
    "nrhip_mf_steps": [p, p, p, p, i64, i32, p, i32, p, f32, f32, f32, p, p, p],
    "nrhip_loss_reduce_steps": [p, i32, i32, i32, f32, p, p],
    "nrhip_lightgcn_step_colshard_fwd": [p, p, p, p, i32, p, p],
    "nrhip_lightgcn_step_colshard_bwd": [p, p, p, p, i32, p, p, f32, f32, f32, f32, p, p],
    "nrhip_lightgcn_partial_dots": [p, p, i32, i32, i32, p, p, p, i32, p, p],
    "nrhip_partials_sum": [p, i32, i32, p, p],
    "nrhip_lightgcn_bpr_grad_given": [p, p, i32, i32, i32, p, p, p, i32, f32, p, p, p, p, p, p, i32, p],
    "nrhip_gemm_workspace_bytes": [i32, i32, i32, p],
    "nrhip_gemm_kmajor": [p, i64, p, i64, i32, i32, i32, p, i64, i32, p, i32, i32, p, sz, p],
    "nrhip_gemm_f32": [p, i64, i32, p, i64, i32, i32, i32, i32, p, i64, i32, p, i32, i32, p, sz, p],
    "nrhip_transpose2d": [p, i64, i32, i32, p, i64, p],
    "nrhip_ngcf_wide_forward": [C.POINTER(NGCFWideBuffers), i32, u64, u64, p],
    "nrhip_ngcf_wide_step": [C.POINTER(NGCFWideBuffers), p, p, p, i32, p, i32, u64, u64, f32, f32, f32, f32, p, p],
    "nrhip_vae_bag_fwd": [p, p, p, i32, i32, p, p, i32, f32, p, u64, u64, p, p, p],
    "nrhip_act_bwd": [p, p, i64, i32, p, p],
    "nrhip_vae_sample": [p, i32, i32, p, f32, u64, u64, p, p, p, p],
    "nrhip_vae_sample_bwd": [p, p, p, i32, i32, f32, p, p],
    "nrhip_vae_softmax_dlogits": [p, i64, i32, i32, p, p, p, p, p],
    "nrhip_vae_dwq0_wide": [p, p, p, i32, i32, p, p, p, p],
    "nrhip_colsum_rows": [p, i64, i32, i32, p, p, sz, p],
    "nrhip_ew_mul": [p, i64, p, i64, i64, i32, p, i64, p],
    "nrhip_ngcf_act_fwd": [p, p, i64, i64, i32, i32, f32, p, i32, u64, u64, i32, p, i64, p, i64, p],
    "nrhip_ngcf_act_bwd": [p, i64, p, i64, p, i64, p, p, i64, p, i64, i32, f32, p, p, p],
    "nrhip_lrelu_drop_fwd": [p, i64, i64, i32, i32, f32, p, i32, u64, u64, i32, i32, p, i64, p, i64, p],
    "nrhip_lrelu_drop_bwd": [p, i64, p, i64, p, i64, p, i64, i32, f32, i32, p, p],
    "nrhip_edge_dropout": [p, i64, f32, p, i32, u64, u64, p, p],
    "nrhip_gather_f32": [p, p, i64, p, p],
    "nrhip_ngcf_mix_bwd": [p, p, i64, p, p, i64, i64, i32, i32, p, p, p],
    "nrhip_route_batch": [p, p, p, i32, i32, i32, i32, i32, p, p, p, p, p, i32, p],
    "nrhip_route_owner_keys": [p, p, i32, p, p, i32, i32, i32, p, p, p],
    "nrhip_sort_u64_segments": [p, p, p, i32, i32, p],
    "nrhip_route_epoch": [p, p, p, i64, i32, i32, i32, i32, i32, i32, p, p, p, p, p, p, p, p],
    "nrhip_route_epoch_owner_keys": [p, p, p, i64, p, p, i32, i32, p, p, p, i32, i32, i64, p, p, p],
    "nrhip_bpr_mf_step_fused": [p, p, p, p, p, p, i32, f32, f32, f32, i32, i32, i32, p, p, p, i32, f32, p, p, p,
                                i32, p, i32, i32, p],
    "nrhip_bpr_mf_fused_flush": [p, p, p, p, p, i32, f32, f32, f32, i32, i64, p],
    "nrhip_adam_sparse_tf_lazy": [p, p, p, p, p, p, i64, i32, p, i32, p, i32, p, i32, i32, f32, f32, f32, p],
    "nrhip_bpr_mf_grad_lazy": [p, p, p, p, p, p, i32, f32, f32, f32, i32, i32, p, p, p, i32, f32, p, p, p,
                               p, p],
    "nrhip_ngcf_workspace_bytes": [i64, psz],
    "nrhip_ngcf_layer_fwd": [p, p, p, p, p, p, i64, i32, f32, p, i32, u64, u64, i32, p, p, i64, p],
    "nrhip_ngcf_layer_bwd": [p, p, p, p, p, p, i64, i32, f32, p, p, i64, p, p, p, p, p, p, p, p, p,
                             p, sz, p],
    "nrhip_spmm_blocked_plan_bytes": [i64, i64, i32, psz],
    "nrhip_spmm_blocked_pack": [p, p, p, p],
    "nrhip_spmm_blocked_plan_create": [p, p, i64, i64, i32, i64, i32, i32, i32, i32, i32, p, sz, p,
                                       C.POINTER(p)],
    "nrhip_spmm_blocked_plan_destroy": [p],
    "nrhip_spmm_blocked_plan_info": [p, p, p, p, p],
    "nrhip_spmm_blocked_tune": [i32],
    "nrhip_spmm_blocked": [p, p, p, p, p, p, p, p, p, p, p],
    "nrhip_spmm_plan_attach_blocked": [p, p, i32],
    "nrhip_spmm_plan_has_blocked": [p, i32],
    "nrhip_spmm_plan_has_wanted": [p, i32],
    "nrhip_spmm_blocked_has_wanted": [p],
    "nrhip_spmm_csr_wanted_layers": [p, p, p, p, i32, p, p, p, p, p, p],
    "nrhip_spmm_blocked_wanted_layers": [p, p, p, p, p, p, p, p, p, p],
    "nrhip_spmm_csr_wanted_batch": [p, p, p, p, i32, p, p, p, p, p, p, p, i32, i32, p, p, p],
    "nrhip_spmm_blocked_wanted_batch": [p, p, p, p, p, p, p, p, p, p, p, i32, i32, p, p, p],
    "nrhip_spmm_csr_adam": [p, p, p, p, i32, p, p, p, p, p, f32, f32, f32, f32, i32, p, p],
    "nrhip_spmm_blocked_adam": [p, p, p, p, p, p, p, p, p, f32, f32, f32, f32, i32, p, p],
    "nrhip_vae_encode": [p, p, p, i32, i32, i32, p, p, p, p, p, p, i32, f32, p, p, f32, u64, u64,
                         p, p, p, p, p, p, p, p, p],
    "nrhip_add_row_bias": [p, i64, i32, i32, p, p],
    "nrhip_vae_workspace_bytes": [i32, i32, psz],
    "nrhip_vae_decoder_loss_grad": [p, i64, i32, i32, i32, p, p, p, p, p, p, p, p, p, p, p, sz, p],
    "nrhip_vae_decoder_fused_workspace_bytes": [i32, i32, psz],
    "nrhip_vae_decoder_fused": [i32, i32, i32, p, p, p, p, p, p, p, p, p, p, p, sz, p, p],
    "nrhip_vae_mid_backward": [i32, i32, i32, i32, f32, p, p, p, p, p, p, p, p, p, p, p, p, p, p,
                               p, p, p, p],
    "nrhip_vae_dwq0": [p, p, p, i32, i32, p, p, p, p],
    "nrhip_axpy": [f32, p, p, i64, p],
    "nrhip_sumsq_accumulate": [p, i64, p, p],
    "nrhip_mean_f32": [p, i32, p, p],
    "nrhip_mean2_f32": [p, p, i32, p, p],
    "nrhip_pairwise_mf_grad": [p, p, i32, i32, p, p, p, i32, f32, i32, p, p, p, p, p, p],
    "nrhip_pointwise_mf_grad": [p, p, i32, i32, p, p, p, i32, f32, i32, p, p, p, p, p, p],
    "nrhip_gather_u8": [p, p, i64, p, p],
    "nrhip_mark_rows": [p, i32, i32, p, p],
    "nrhip_sort_u64": [p, i32, p],
    "nrhip_rows_sum_sorted": [p, i32, p, i32, p, i64, p, p],
    "nrhip_rows_sum_sorted2": [p, i32, p, i32, p, i64, p, p, i64, p, p],
    "nrhip_optimizer_rows_tf": [i32, p, p, p, p, p, i64, i32, f32, f32, f32, f32, p],
    "nrhip_optimizer_dense_tf": [i32, p, p, p, p, i64, f32, f32, f32, f32, i32, p],
    "nrhip_rows_gather": [p, i32, i32, p, p, i64, p],
    "nrhip_rows_gather_ld": [p, i32, i32, p, i64, p, i64, p],
    "nrhip_rows_gather2": [p, i32, i32, p, i64, p, i64, p, i64, p, i64, p],
    "nrhip_rows_scatter_add": [p, i32, i32, p, i64, p, p],
    "nrhip_scale": [p, f32, p, i64, p],
    "nrhip_add": [p, p, p, i64, p],
    "nrhip_div_scalar": [p, f32, p, i64, p],
    "nrhip_add2d": [p, i64, p, i64, p, i64, i64, i32, p],
    "nrhip_copy2d": [p, i64, p, i64, i64, i32, p],
}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here == ABI mismatch: fail loudly
    _fn.argtypes = _args
    _fn.restype = C.c_int
lib.nrhip_abi_version.argtypes = []
lib.nrhip_abi_version.restype = C.c_int
lib.nrhip_last_error.argtypes = []
lib.nrhip_last_error.restype = C.c_char_p

EXPORTED = sorted(list(SIGNATURES) + ["nrhip_abi_version", "nrhip_last_error"])

ABI_VERSION = 4      # 2: deterministic row-gradient sums (batch plans); 3: one-launch BPR-MF step (mf_buffers.tw);
                     # 4: nrhip_mf_steps takes a per-step terms buffer (one loss reduction per call)
if lib.nrhip_abi_version() != ABI_VERSION:  # pragma: no cover
    raise ImportError("libneurec_hip.so ABI version %d, expected %d"
                      % (lib.nrhip_abi_version(), ABI_VERSION))


def last_error():
    return lib.nrhip_last_error().decode("utf-8", "replace")


def check(rc):
    """Translate a status code into the exception type the reference would raise."""
    if rc == 0:
        return
    msg = last_error()
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise NeuRecHipError("libneurec_hip status %d: %s" % (rc, msg))


def call(name, *args):
    check(getattr(lib, name)(*args))


def is_stale():
    """True when the .so on disk was built from different sources than the tree holds."""
    return not _build.is_current()
