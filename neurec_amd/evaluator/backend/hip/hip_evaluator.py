"""HIPEvaluator — `eval_score_matrix` with the signature of the reference's CPPEvaluator
(evaluator/backend/cpp/cpp_evaluator.pyx:28-42), computed by the HIP selection/metric kernels."""
import numpy as np

from ...abstract_evaluator import AbstractEvaluator

float_type = np.float32


class HIPEvaluator(AbstractEvaluator):
    def __init__(self):
        super(HIPEvaluator, self).__init__()

    def eval_score_matrix(self, score_matrix, test_items, metric, top_k, thread_num=None):
        """score_matrix: [B, N] float32 (host ndarray or device tensor); test_items: B lists of
        ground-truth column ids; metric: list of ids 1..5; returns float32 ndarray
        [B, len(metric)*top_k], metric-major (evaluate.h:43-47).  `thread_num` is accepted for
        signature compatibility; the launch shape is one wave per row."""
        import torch
        from .... import engine as E
        dev = E.require_gpu()
        if isinstance(score_matrix, torch.Tensor):
            scores = score_matrix.to(device=dev, dtype=torch.float32)
            if scores.stride(-1) != 1:
                scores = scores.contiguous()
        else:
            scores = torch.from_numpy(np.ascontiguousarray(score_matrix, dtype=float_type)).to(dev)
        rows, cols = scores.shape
        if len(test_items) != rows:
            raise ValueError("len(test_items) must equal the number of score rows")
        ptr = np.zeros(rows + 1, dtype=np.int64)
        ptr[1:] = np.cumsum([len(t) for t in test_items])
        idx = np.zeros(int(ptr[-1]), dtype=np.int32)
        for r, t in enumerate(test_items):
            idx[ptr[r]:ptr[r + 1]] = np.sort(np.fromiter(t, dtype=np.int64, count=len(t)))
        truth = E.DeviceCSR(ptr, idx, cols)
        out = E.eval_scores(scores, truth, [int(m) for m in metric], int(top_k))
        return out.cpu().numpy()
