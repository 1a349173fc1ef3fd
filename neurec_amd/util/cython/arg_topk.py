"""`arg_topk(ranking_scores, top_k=50, thread_num=None)` of util/cython/arg_topk.pyx:16-35 on
the HIP selection kernel (same std::partial_sort_copy tie order)."""
import numpy as np

from .tools import float_type


def arg_topk(ranking_scores, top_k=50, thread_num=None):
    import torch
    from ... import engine as E
    if isinstance(ranking_scores, torch.Tensor):
        scores = ranking_scores.to(device=E.require_gpu(), dtype=torch.float32).contiguous()
    else:
        scores = torch.from_numpy(np.ascontiguousarray(ranking_scores, dtype=float_type)).to(E.require_gpu())
    if scores.dim() != 2:
        raise ValueError("ranking_scores must be 2-D")
    return E.arg_topk(scores, int(top_k)).cpu().numpy()
