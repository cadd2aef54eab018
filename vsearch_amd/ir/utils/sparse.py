"""Sparsification helpers with the reference's names and argument meaning
(/root/reference/src/ir/utils/sparse.py:6-29), computed by the HIP kernels of libvsearch_hip.so.

Inputs may be torch tensors (CPU or CUDA) or numpy arrays; compute always happens on the MI355X
(`device` = the tensor's CUDA device, else GPU 0) and results come back on the input's device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _native as nat
from ...device_index import current_stream


def _dev_of(t: torch.Tensor) -> int:
    return t.device.index if t.is_cuda and t.device.index is not None else (torch.cuda.current_device() if t.is_cuda else 0)


def _stage(t: torch.Tensor, dtype) -> tuple[torch.Tensor, int]:
    """-> contiguous tensor of `dtype` on a CUDA device (GPU 0 for host inputs), device ordinal."""
    nat.require_device()
    dev = _dev_of(t)
    return t.detach().to(device=torch.device("cuda", dev), dtype=dtype).contiguous(), dev


def elu1p(x):
    """F.elu(x) + 1 (sparse.py:6)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    src_device = x.device
    xd, dev = _stage(x, torch.float32)
    out = torch.empty_like(xd)
    nat.check(nat.lib().vs_elu1p(C.c_void_p(xd.data_ptr()), xd.numel(), C.c_void_p(out.data_ptr()), dev, current_stream(dev)))
    return out.to(src_device)


def build_topk_mask(embs, k: int = 768, dim: int = -1):
    """Bool mask of the k largest entries of each row (sparse.py:8-14). Ties at the k-th value go to
    the lowest column ids (torch.topk leaves them unspecified)."""
    if isinstance(embs, np.ndarray):
        embs = torch.from_numpy(embs)
    if dim not in (-1, embs.dim() - 1):
        raise NotImplementedError("build_topk_mask: only the last dimension is supported")
    src_device, shape = embs.device, embs.shape
    xd, dev = _stage(embs.reshape(-1, shape[-1]), torch.float32)
    B, V = xd.shape
    mask = torch.empty((B, V), dtype=torch.uint8, device=xd.device)
    nat.check(nat.lib().vs_topk_mask(C.c_void_p(xd.data_ptr()), B, V, V, int(k), C.c_void_p(mask.data_ptr()), dev, current_stream(dev)))
    return mask.bool().reshape(shape).to(src_device)


def topk_sparsify(emb_dense: torch.Tensor, k: int, dim: int = -1):
    """emb * build_topk_mask(emb, k) (sparse.py:16-19)."""
    return emb_dense * build_topk_mask(emb_dense, k=k, dim=dim)


def build_bow_mask(text_ids, vocab_size=30522, shift_num=0, norm=False):
    """Multi-hot of token ids over the vocabulary, first `shift_num` columns dropped, optional L2 row
    normalisation (sparse.py:21-29). text_ids: int64 [N, L] -> float32 [N, vocab_size - shift_num]."""
    if isinstance(text_ids, np.ndarray):
        text_ids = torch.from_numpy(text_ids)
    src_device = text_ids.device
    ids, dev = _stage(text_ids, torch.int64)
    N, L = ids.shape
    out = torch.empty((N, vocab_size - shift_num), dtype=torch.float32, device=ids.device)
    try:
        nat.check(nat.lib().vs_bow_mask(C.c_void_p(ids.data_ptr()), N, L, int(vocab_size), int(shift_num), int(bool(norm)),
                                        C.c_void_p(out.data_ptr()), dev, current_stream(dev)))
    except ValueError as e:            # scatter_ raises RuntimeError for out-of-range ids
        raise RuntimeError(str(e)) from None
    return out.to(src_device)


def apply_embed_mask_(emb: torch.Tensor, input_ids, vocab_size: int, shift_num: int, topk, activate_lexical: bool, bow: bool = False):
    """In-place mask stage of VDREncoder.embed (vdr.py:152-169) on a CUDA tensor emb [B, V]."""
    assert emb.is_cuda and emb.dtype == torch.float32 and emb.is_contiguous()
    dev = _dev_of(emb)
    B, V = emb.shape
    ids = None
    L = 0
    if input_ids is not None:
        ids = input_ids.detach().to(device=emb.device, dtype=torch.int64).contiguous()
        L = ids.shape[1]
    tk = -1 if topk is None else int(topk)
    nat.check(nat.lib().vs_embed_mask(C.c_void_p(emb.data_ptr()), V, C.c_void_p(ids.data_ptr()) if ids is not None else None,
                                      B, L, int(vocab_size), int(shift_num), tk, int(bool(activate_lexical)), int(bool(bow)),
                                      dev, current_stream(dev)))
    return emb


FUSED_CSR_MAX_KEPT = 8192      # kMrStage (csrc/mask_rows_fast.h): kept elements of a row the fused kernel stages


def embed_mask_to_csr(emb: torch.Tensor, input_ids, vocab_size: int, shift_num: int, topk: int, activate_lexical: bool):
    """The mask stage of VDREncoder.embed (vdr.py:152-169) fused with Tensor.to_sparse_csr() (retriever.py:304): pooled activations
    emb [B, V] (CUDA fp32, NOT modified) -> (rowptr int64 [B+1], cols int32 [nnz], vals fp32 [nnz]) of emb * (topk_mask | lexical_mask).
    One read of [B, V]; the masked dense batch is never written.  NotImplementedError outside the fused kernel's range (topk <= 0,
    V > 32 Ki, topk + L > 8192): callers then use apply_embed_mask_ + dense_to_csr."""
    assert emb.is_cuda and emb.dtype == torch.float32 and emb.is_contiguous()
    dev = _dev_of(emb)
    B, V = emb.shape
    ids, L = None, 0
    if activate_lexical:
        ids = input_ids.detach().to(device=emb.device, dtype=torch.int64).contiguous()
        L = ids.shape[1]
    tk = -1 if topk is None else int(topk)
    if tk <= 0 or V > 32768 or min(V, tk + L) > FUSED_CSR_MAX_KEPT:
        raise NotImplementedError("embed_mask_to_csr serves top-k masks of V <= 32 Ki columns and topk + L <= 8192 kept elements a row")
    cap = B * min(V, tk + L)
    rowptr = torch.empty(B + 1, dtype=torch.int64, device=emb.device)
    cols = torch.empty(max(cap, 1), dtype=torch.int32, device=emb.device)
    vals = torch.empty(max(cap, 1), dtype=torch.float32, device=emb.device)
    nat.check(nat.lib().vs_embed_mask_to_csr(C.c_void_p(emb.data_ptr()), V, C.c_void_p(ids.data_ptr()) if ids is not None else None, B, L,
                                             int(vocab_size), int(shift_num), tk, int(bool(activate_lexical)), C.c_void_p(rowptr.data_ptr()),
                                             C.c_void_p(cols.data_ptr()), C.c_void_p(vals.data_ptr()), cap, dev, current_stream(dev)))
    nnz = int(rowptr[-1].item())
    return rowptr, cols[:nnz], vals[:nnz]


def head_pool(logits: torch.Tensor):
    """elu1p then max over the sequence axis of [B, L, V] logits (vdr.py:73-75) -> [B, V]."""
    assert logits.is_cuda and logits.dim() == 3
    x = logits.detach().to(torch.float32).contiguous()
    dev = _dev_of(x)
    B, L, V = x.shape
    out = torch.empty((B, V), dtype=torch.float32, device=x.device)
    nat.check(nat.lib().vs_head_pool(C.c_void_p(x.data_ptr()), B, L, V, C.c_void_p(out.data_ptr()), dev, current_stream(dev)))
    return out


def head_pool_mean_topk(logits: torch.Tensor, topk: int):
    """Mean of the `topk` largest elu1p(logits) over the sequence axis of [B, L, V] logits (vdr.py:76-79) -> [B, V]."""
    assert logits.is_cuda and logits.dim() == 3
    x = logits.detach().to(torch.float32).contiguous()
    dev = _dev_of(x)
    B, L, V = x.shape
    out = torch.empty((B, V), dtype=torch.float32, device=x.device)
    # (pooling_topk > L -> RuntimeError "selected index k out of range", like torch.topk)
    nat.check(nat.lib().vs_head_pool_mean_topk(C.c_void_p(x.data_ptr()), B, L, V, int(topk), C.c_void_p(out.data_ptr()), dev, current_stream(dev)))
    return out


def head_project_pool(hidden_ln: torch.Tensor, weight: torch.Tensor):
    """Fused encoder head (vdr.py:72-75): elu1p(max over positions of hidden_ln @ weight.T) without the [B, L, V] logits.
    hidden_ln [B, L, H] and weight [V, H]: CUDA fp32, H % 32 == 0."""
    assert hidden_ln.is_cuda and weight.is_cuda and hidden_ln.dim() == 3 and weight.dim() == 2
    h = hidden_ln.detach().to(torch.float32).contiguous()
    w = weight.detach().to(torch.float32).contiguous()
    dev = _dev_of(h)
    B, L, H = h.shape
    V = w.shape[0]
    out = torch.empty((B, V), dtype=torch.float32, device=h.device)
    nat.check(nat.lib().vs_head_project_pool(C.c_void_p(h.data_ptr()), C.c_void_p(w.data_ptr()), B, L, H, V, C.c_void_p(out.data_ptr()),
                                             dev, current_stream(dev)))
    return out


def dense_to_csr(x: torch.Tensor):
    """Tensor.to_sparse_csr() (retriever.py:304) for a CUDA fp32 [B, V] tensor -> (rowptr int64 [B+1],
    cols int32 [nnz], vals fp32 [nnz]) CUDA tensors."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
    x = x.contiguous()
    dev = _dev_of(x)
    B, V = x.shape
    rowptr = torch.empty(B + 1, dtype=torch.int64, device=x.device)
    s = current_stream(dev)
    nat.check(nat.lib().vs_dense_to_csr(C.c_void_p(x.data_ptr()), B, V, V, C.c_void_p(rowptr.data_ptr()), None, None, 0, dev, s))
    nnz = int(rowptr[-1].item())
    cols = torch.empty(max(nnz, 1), dtype=torch.int32, device=x.device)
    vals = torch.empty(max(nnz, 1), dtype=torch.float32, device=x.device)
    nat.check(nat.lib().vs_dense_to_csr(C.c_void_p(x.data_ptr()), B, V, V, C.c_void_p(rowptr.data_ptr()), C.c_void_p(cols.data_ptr()),
                                        C.c_void_p(vals.data_ptr()), nnz, dev, s))
    return rowptr, cols[:nnz], vals[:nnz]
