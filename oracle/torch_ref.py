"""The reference's ``Index.search`` restated call-for-call (TEST INFRASTRUCTURE / CPU baseline).

/root/reference/src/ir/retriever/index.py:88-94:
    q_embs = q_embs.to(self.device).type(self.vector.dtype)      # :89
    scores = torch.matmul(q_embs, self.vector.t())               # :91  (CSR -> MKL sparse addmm on CPU)
    scores_topk = scores.topk(k)                                 # :92
Reference files never travel to the GPU box, so ``bench.py``'s ``cpu_baseline`` leg times THIS
restatement (kind = "port") on the box's host cores.  It is pinned to the reference by
tests/test_oracle_golden.py (same ids/scores as the goldens modulo tie order).
"""
from __future__ import annotations

import warnings

import numpy as np
import torch


def make_csr(indptr, indices, data, shape):
    """index.py:144-161 (_scipy_csr_to_torch_csr): int64 crow/col indices, values as given."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.sparse_csr_tensor(torch.from_numpy(np.ascontiguousarray(indptr, dtype=np.int64)),
                                       torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int64)),
                                       torch.from_numpy(np.ascontiguousarray(data)), size=tuple(shape))


def search(vector: torch.Tensor, q_embs: torch.Tensor, k: int):
    q_embs = q_embs.to(vector.device).type(vector.dtype)
    with torch.no_grad():
        scores = torch.matmul(q_embs, vector.t())
    scores_topk = scores.topk(k)
    return scores_topk.indices, scores_topk.values
