"""Shared host code of the HIP-backed general recommenders."""
import numpy as np


def predict_scores(user_table, item_table, user_ids, candidate_items=None):
    """`predict` contract of the reference models (MF.py:120-134): a [B, I] float32 array, or
    — in candidate mode — a list of per-user score arrays.  The scores are computed by the
    fp32-MFMA scoring kernel and copied to the host (this entrance exists for plugin
    compatibility; the evaluator itself uses the on-device factor path)."""
    import torch
    from ... import engine as E
    users = torch.tensor(np.asarray(list(user_ids), dtype=np.int32), device=user_table.device)
    gemm = E.score_gemm_for(item_table, max(users.numel(), 1))
    ratings = gemm(user_table, users).cpu().numpy()[:, :item_table.shape[0]]
    if candidate_items is not None:
        return [rating[items] for rating, items in zip(ratings, candidate_items)]
    return np.ascontiguousarray(ratings)
