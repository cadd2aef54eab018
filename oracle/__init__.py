"""CPU oracle for the vocabulary-space retrieval path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  ``vsearch_amd`` never imports it (tests/test_boundary.py greps
for that).  The restatement is pinned against golden vectors captured from the reference itself
(``tools/gen_golden.py`` -> ``tests/golden/*.npz``, verified by ``tests/test_oracle_golden.py``).

Contents
  vs_oracle.c   plain-C restatement (each function cites the reference file:line it follows)
  torch_ref.py  the reference's three torch calls for ``Index.search`` restated verbatim
                (src/ir/retriever/index.py:89-92) -- the timed CPU baseline ("port")
  compare.py    tie-/near-tie-aware comparison of top-k results (SURVEY.md §8(c) comparator)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libvs_oracle.so")
    src = os.path.join(_HERE, "vs_oracle.c")
    # build only when asked or missing: a snapshot copy can reorder mtimes, and several bench ranks must
    # not race on `make` (the driver's build() compiles it up front)
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvs_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def csr_search(indptr, indices, data, n_cols, q, k, acc64=False, return_all=False):
    """SparseIndex.search restated (index.py:88-94). data=None -> binary (BoT) index."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    data = _f32(data) if data is not None else None
    q = _f32(q)
    B, n = q.shape[0], len(indptr) - 1
    assert q.shape[1] == n_cols
    ids = np.empty((B, k), dtype=np.int64)
    scores = np.empty((B, k), dtype=np.float32)
    allsc = np.empty((B, n), dtype=np.float32) if return_all else None
    rc = lib().vso_csr_search(_p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_float),
                              C.c_int64(n), C.c_int32(n_cols), _p(q, C.c_float), C.c_int32(B), C.c_int64(k),
                              C.c_int(int(acc64)), _p(ids, C.c_int64), _p(scores, C.c_float), _p(allsc, C.c_float))
    if rc != 0:
        raise RuntimeError("selected index k out of range")       # torch.topk's error (index.py:92)
    return (ids, scores, allsc) if return_all else (ids, scores)


def dense_search(mat, q, k, acc64=False):
    mat, q = _f32(mat), _f32(q)
    B, n = q.shape[0], mat.shape[0]
    ids = np.empty((B, k), dtype=np.int64)
    scores = np.empty((B, k), dtype=np.float32)
    rc = lib().vso_dense_search(_p(mat, C.c_float), C.c_int64(n), C.c_int32(mat.shape[1]), _p(q, C.c_float),
                                C.c_int32(B), C.c_int64(k), C.c_int(int(acc64)), _p(ids, C.c_int64), _p(scores, C.c_float))
    if rc != 0:
        raise RuntimeError("selected index k out of range")
    return ids, scores


def merge_topk(cand_ids, cand_scores, k):
    cand_ids = np.ascontiguousarray(cand_ids, dtype=np.int64)
    cand_scores = _f32(cand_scores)
    B, n = cand_ids.shape
    ids = np.empty((B, k), dtype=np.int64)
    scores = np.empty((B, k), dtype=np.float32)
    rc = lib().vso_merge_topk(_p(cand_ids, C.c_int64), _p(cand_scores, C.c_float), C.c_int32(B), C.c_int64(n),
                              C.c_int64(k), _p(ids, C.c_int64), _p(scores, C.c_float))
    if rc != 0:
        raise RuntimeError("k > number of candidates")
    return ids, scores


def elu1p(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().vso_elu1p(_p(x, C.c_float), C.c_int64(x.size), _p(out, C.c_float))
    return out


def topk_mask(x, k):
    x = _f32(x)
    B, V = x.shape
    mask = np.empty((B, V), dtype=np.uint8)
    if lib().vso_topk_mask(_p(x, C.c_float), C.c_int32(B), C.c_int32(V), C.c_int32(k), _p(mask, C.c_uint8)) != 0:
        raise RuntimeError("selected index k out of range")
    return mask.astype(bool)


def bow_mask(ids, vocab=30522, shift=0, norm=False):
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    B, L = ids.shape
    out = np.empty((B, vocab - shift), dtype=np.float32)
    if lib().vso_bow_mask(_p(ids, C.c_int64), C.c_int32(B), C.c_int32(L), C.c_int32(vocab), C.c_int32(shift),
                          C.c_int(int(norm)), _p(out, C.c_float)) != 0:
        raise RuntimeError("index out of range in scatter_")
    return out


def embed_mask(emb, ids, vocab=30522, shift=999, topk=768, activate_lexical=True, bow=False):
    emb = _f32(emb).copy()
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    B, L = ids.shape
    tk = -1 if topk is None else int(topk)
    if lib().vso_embed_mask(_p(emb, C.c_float), _p(ids, C.c_int64), C.c_int32(B), C.c_int32(L), C.c_int32(vocab),
                            C.c_int32(shift), C.c_int32(tk), C.c_int(int(activate_lexical)), C.c_int(int(bow))) != 0:
        raise RuntimeError("embed_mask failed")
    return emb


def head_pool(logits):
    logits = _f32(logits)
    B, L, V = logits.shape
    out = np.empty((B, V), dtype=np.float32)
    lib().vso_head_pool(_p(logits, C.c_float), C.c_int32(B), C.c_int32(L), C.c_int32(V), _p(out, C.c_float))
    return out


def bot_build(token_lists, vocab=30522, shift=999, max_token=None):
    """_build_bot_vectors restated (retriever.py:208-253). token_lists: list of int lists (already truncated)."""
    offsets = np.zeros(len(token_lists) + 1, dtype=np.int64)
    np.cumsum([len(t) for t in token_lists], out=offsets[1:])
    tokens = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32) for t in token_lists])
                                  if len(token_lists) else np.zeros(0, np.int32))
    n = len(token_lists)
    indptr = np.empty(n + 1, dtype=np.int64)
    mt = int(max_token) if max_token else 0
    args = (_p(tokens, C.c_int32), _p(offsets, C.c_int64), C.c_int64(n), C.c_int32(vocab), C.c_int32(shift), C.c_int32(mt))
    if lib().vso_bot_build(*args, _p(indptr, C.c_int64), None) != 0:
        raise IndexError("token id out of range")
    indices = np.empty(int(indptr[-1]), dtype=np.int32)
    lib().vso_bot_build(*args, _p(indptr, C.c_int64), _p(indices, C.c_int32))
    return indptr, indices


def synth_csr(seed, row0, n_rows, n_cols=29523, nnz=768, kind=0, val_law=0):
    """Fast twin of vsearch_amd.synth.synth_csr."""
    indptr = np.empty(n_rows + 1, dtype=np.int64)
    a = (C.c_uint64(seed), C.c_int64(row0), C.c_int64(n_rows), C.c_int32(n_cols), C.c_int32(nnz), C.c_int(kind), C.c_int(val_law))
    if lib().vso_synth_csr(*a, _p(indptr, C.c_int64), None, None) != 0:
        raise ValueError("bad synth params")
    indices = np.empty(int(indptr[-1]), dtype=np.int32)
    data = np.empty(int(indptr[-1]), dtype=np.float32)
    lib().vso_synth_csr(*a, _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_float))
    return indptr, indices, data


def synth_queries(seed, n_q, n_cols=29523, nnz_q=776, val_law=0, q0=0, kind=0):
    indptr, cols, data = synth_csr(seed, q0, n_q, n_cols, nnz_q, kind, val_law)
    q = np.zeros((n_q, n_cols), dtype=np.float32)
    q[np.repeat(np.arange(n_q), np.diff(indptr)), cols] = data
    return q
