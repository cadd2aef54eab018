"""Retriever with the reference's interface (/root/reference/src/ir/retriever/retriever.py:20-348):
``retrieve / process_query / build_index / save_index / load_index`` with the same signatures, on
top of the device-resident indexes of ``index.py``.

Deviations (supersets): ``retrieve(index=...)`` really uses the passed index;
``load_index`` accepts ``IndexType`` as well as ``str``; ``build_index(SPARSE)`` emits CSR batch by
batch on the GPU instead of materialising the dense [N, V] matrix; the bag-of-token builder
implements the intended per-document semantics for any batch size (upstream aliases batches).
Training-time methods (``forward`` with negatives, ``retireve_negatives``: hard-negative mining for the training loop,
retriever.py:150-205) are out of scope and absent.
"""
from __future__ import annotations

import logging
from typing import List, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor as T

from ... import _native as nat
from ..biencoder.biencoder import BiEncoder, BiEncoderConfig
from ..utils import sparse as sp
from .index import BoTIndex, Index, IndexType, SearchResults, SparseIndex

logger = logging.getLogger(__name__)


class RetrieverConfig(BiEncoderConfig):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)


class Retriever(BiEncoder):
    config_class = RetrieverConfig

    def __init__(self, config: RetrieverConfig, index: Index = None, **kwargs):
        super().__init__(config, **kwargs)
        self.config = config
        self.index = index

    # ---- queries (retriever.py:74-104) -------------------------------------------------------------
    def process_query(self, queries: Union[str, List[str], np.ndarray, T], dropout: float = 0, a: int = None,
                      batch_size: int = 32) -> T:
        """str / list of str -> encoder_q.embed(topk=a); ndarray -> tensor; tensor -> as is; optional dropout."""
        active = a or self.encoder_q.config.topk
        if isinstance(queries, str):
            q_emb = self.encoder_q.embed([queries], batch_size=batch_size, topk=active)
        elif isinstance(queries, list) and len(queries) > 0 and isinstance(queries[0], str):
            q_emb = self.encoder_q.embed(queries, batch_size=batch_size, topk=active)
        elif isinstance(queries, np.ndarray):
            q_emb = torch.Tensor(queries)
        elif isinstance(queries, T):
            q_emb = queries
        else:
            raise NotImplementedError(f"Query type {type(queries)} not supported")
        if dropout:
            q_emb = F.dropout(q_emb, p=dropout)
        return q_emb

    # ---- retrieval (retriever.py:107-148) ----------------------------------------------------------
    def retrieve(self, queries: Union[List[str], np.ndarray, T], k: int = 5, dropout: float = 0, a: int = None,
                 index: Index = None, rerank: bool = False, batch_size: int = 32) -> SearchResults:
        index = index or self.index
        if index is None:
            raise RuntimeError("no index: call build_index / load_index first")
        a = a or self.encoder_q.config.topk
        q_emb = self.process_query(queries, dropout, a, batch_size=batch_size)
        results = index.search(q_emb, k=k)
        if rerank and index.index_type == IndexType.BAG_OF_TOKEN:
            results = self._rerank(index, q_emb, results, k, batch_size)
        return results

    def _rerank(self, index: Index, q_emb: T, results: SearchResults, k: int, batch_size: int) -> SearchResults:
        """Re-embed the k hits with encoder_p, score against q, re-sort (retriever.py:137-147) -- on the device:
        ``vs_rerank_scores`` per re-embedding batch (the reference's dense [B*k, V] tensor is never held) and
        ``vs_rerank_topk`` for the order (score descending; equal scores keep their first-stage order)."""
        import ctypes as C
        from ... import _native as nat
        from ...device_index import current_stream
        hit_ids = results.ids
        B = int(hit_ids.shape[0])
        texts = [index.get_sample(i) for i in hit_ids.flatten().tolist()]
        nat.require_device()
        dev = hit_ids.device if hit_ids.is_cuda else torch.device("cuda", 0)
        ordinal = dev.index or 0
        stream = current_stream(ordinal)
        q = q_emb.to(dev).to(torch.float32).contiguous()
        ids_dev = hit_ids.to(dev).contiguous()
        scores = torch.empty((B, k), dtype=torch.float32, device=dev)
        chunk = max(int(batch_size), 1) * 32               # re-embedded passages held at a time
        for r0 in range(0, B * k, chunk):
            p_emb = self.encoder_p.embed(texts[r0:r0 + chunk], batch_size=batch_size, require_grad=False)
            if p_emb.dtype not in (torch.float32, torch.float16):
                p_emb = p_emb.to(torch.float32)
            p_emb = p_emb.to(dev).contiguous()
            nat.check(nat.lib().vs_rerank_scores(C.c_void_p(p_emb.data_ptr()), nat.VS_F16 if p_emb.dtype == torch.float16 else nat.VS_F32,
                                                 int(p_emb.shape[1]), int(p_emb.shape[0]), r0, C.c_void_p(q.data_ptr()), int(q.shape[1]), B, int(k),
                                                 int(q.shape[1]), C.c_void_p(scores.data_ptr()), ordinal, stream))
        out_ids = torch.empty_like(ids_dev)
        out_scores = torch.empty_like(scores)
        nat.check(nat.lib().vs_rerank_topk(C.c_void_p(scores.data_ptr()), C.c_void_p(ids_dev.data_ptr()), B, int(k), C.c_void_p(out_ids.data_ptr()),
                                           C.c_void_p(out_scores.data_ptr()), ordinal, stream))
        return SearchResults(out_ids, out_scores)

    def retireve_negatives(self, *args, **kwargs):
        """Hard-negative mining for the reference's TRAINING loop (retriever.py:150-205, called from forward() at :51): out of this
        build's scope (the retrieval hot path, no training).  Present so that a caller gets a clear message, not an
        AttributeError."""
        raise NotImplementedError("Retriever.retireve_negatives (hard-negative mining for training, reference retriever.py:150-205) is not part of "
                                  "the MI355X retrieval path; use retrieve() and filter the hits")

    retrieve_negatives = retireve_negatives

    # ---- index build (retriever.py:208-317) ----------------------------------------------------------
    def _tokenize_for_bot(self, texts: List[str], max_len: int):
        return self.encoder_p.tokenizer(texts, max_length=max_len, truncation=True)["input_ids"]

    def _build_bot_vectors(self, texts: List[str], batch_size: int = 32, max_len: int = 128, max_token: int = None,
                           num_shift: int = 999, fp16: int = True):
        """Binary bag-of-token CSR of the corpus (retriever.py:208-253): per document the set of token ids
        >= num_shift (optionally only the first `max_token` distinct ids, [CLS] included in the count)."""
        import ctypes as C
        vocab = len(self.encoder_p.tokenizer.vocab)
        token_lists = []
        for s in range(0, len(texts), batch_size):
            token_lists.extend(self._tokenize_for_bot(texts[s:s + batch_size], max_len))
        offsets = np.zeros(len(token_lists) + 1, dtype=np.int64)
        np.cumsum([len(t) for t in token_lists], out=offsets[1:])
        tokens = (np.concatenate([np.asarray(t, dtype=np.int32) for t in token_lists]) if token_lists else np.zeros(0, np.int32))
        tokens = np.ascontiguousarray(tokens)
        n = len(token_lists)
        indptr = np.empty(n + 1, dtype=np.int64)
        args = (C.c_void_p(tokens.ctypes.data), C.c_void_p(offsets.ctypes.data), n, int(vocab), int(num_shift), int(max_token or 0))
        try:
            nat.check(nat.lib().vs_bot_build(*args, C.c_void_p(indptr.ctypes.data), None))
        except ValueError as e:
            raise IndexError(str(e)) from None
        indices = np.empty(max(int(indptr[-1]), 1), dtype=np.int32)
        nat.check(nat.lib().vs_bot_build(*args, C.c_void_p(indptr.ctypes.data), C.c_void_p(indices.ctypes.data)))
        indices = indices[:int(indptr[-1])]
        values = torch.ones(indices.shape[0], dtype=torch.float16 if fp16 else torch.float32)
        return torch.sparse_csr_tensor(torch.from_numpy(indptr), torch.from_numpy(indices.astype(np.int64)), values,
                                       size=(n, vocab - num_shift))

    def _build_embedding_vectors(self, texts: List[str], batch_size: int = 32, max_len: int = 128, num_shift: int = 0) -> T:
        """Dense [N, V] passage embeddings (retriever.py:256-282)."""
        parts = []
        for s in range(0, len(texts), batch_size):
            emb = self.encode_corpus(texts[s:s + batch_size], batch_size=batch_size, max_len=max_len, convert_to_tensor=True)
            parts.append(emb[:, num_shift:])
        return torch.cat(parts, dim=0)

    def _build_embedding_csr(self, texts: List[str], batch_size: int = 32, max_len: int = 128):
        """Same embeddings as `_build_embedding_vectors(...).to_sparse_csr()` (retriever.py:303-304), emitted as CSR per batch
        on the GPU and appended to the index IN HBM (``vs_index_append_csr`` takes the device pointers): neither the dense
        [N, V] fp32 matrix (118 KB per passage) nor a host copy of the CSR pieces is ever held.  -> DeviceIndex."""
        from ...device_index import DeviceIndex
        from ... import _native as nat
        n = len(texts)
        dev_index, row_cap, pk_cap = None, n, 0
        pending = []                                        # batches embedded before the first reservation / beyond it
        # the mask stage and to_sparse_csr() fused (encoder.embed_csr -> vs_embed_mask_to_csr: the masked dense batch is never written) when
        # the encoder offers it; else embed + vs_dense_to_csr
        fused = hasattr(self.encoder_p, "embed_csr") and type(self).encode_corpus is BiEncoder.encode_corpus
        batches = self.encode_corpus_csr(texts, batch_size=batch_size, max_len=max_len) if fused else None
        for s in range(0, n, batch_size):
            if fused:
                try:
                    rp, ci, va, n_cols = next(batches)
                except (nat.VsearchNativeError, ValueError):
                    # (a native error of the fused kernel -- raised here, outside the append below: the same host fallback serves it)
                    if dev_index is not None:
                        dev_index.close()
                    return None
            else:
                emb = self.encode_corpus(texts[s:s + batch_size], batch_size=batch_size, max_len=max_len, convert_to_tensor=True)
                rp, ci, va = sp.dense_to_csr(emb.float().contiguous())        # device tensors
                n_cols = int(emb.shape[1])
            if dev_index is None:
                # reserve from the first batch: the encoder keeps at most topk (+ the lexical tokens) non-zeros per passage
                per_row = int((rp[1:] - rp[:-1]).max().item()) if rp.numel() > 1 else 0
                bound = max(per_row, int(getattr(getattr(self.encoder_p, "config", None), "topk", 0) or 0)) + int(max_len)
                pk_cap = n * ((min(bound, n_cols) + 7) // 8)
                dev_index = DeviceIndex.reserved(row_cap, max(pk_cap, 1), n_cols, nat.VS_F32, device=rp.device.index or 0)
            try:
                dev_index.append_csr(rp, ci, va)
            except (nat.VsearchNativeError, ValueError):
                pending.append((rp.cpu(), ci.cpu(), va.cpu(), s))          # a passage denser than the bound: rebuild with exact sizes
                break
        if pending:
            return None
        return dev_index

    def _build_embedding_csr_host(self, texts: List[str], batch_size: int = 32, max_len: int = 128):
        """Fallback of `_build_embedding_csr` (a passage denser than the reservation bound): CSR pieces gathered on the host."""
        ptrs, cols, vals, base, V = [torch.zeros(1, dtype=torch.int64)], [], [], 0, None
        for s in range(0, len(texts), batch_size):
            emb = self.encode_corpus(texts[s:s + batch_size], batch_size=batch_size, max_len=max_len, convert_to_tensor=True)
            V = emb.shape[1]
            rp, ci, va = sp.dense_to_csr(emb.float().contiguous())
            ptrs.append(rp[1:].cpu() + base)
            base += int(rp[-1].item())
            cols.append(ci.cpu())
            vals.append(va.cpu())
        return torch.sparse_csr_tensor(torch.cat(ptrs), torch.cat(cols).to(torch.int64), torch.cat(vals), size=(len(texts), V))

    def build_index(self, texts: List[str], batch_size=32, index_type=IndexType.DENSE, bag_of_token=False, devices=None):
        """retriever.py:284-317.  `devices` (not in the reference): deal the built sparse / bag-of-token index over several GPUs of this
        process in contiguous row ranges (SparseIndex.shard_rows) -- None, "all", a count, or a list of GPU ordinals."""
        if isinstance(index_type, str):
            index_type = IndexType(index_type.lower())
        elif not isinstance(index_type, IndexType):
            raise TypeError("index_type must be an instance of IndexType, int, or str.")
        self.index_type = index_type
        if index_type == IndexType.DENSE:
            self.index = Index()
            self.index.data = texts
            self.index.vector = self._build_embedding_vectors(texts, batch_size=batch_size)
        elif index_type == IndexType.SPARSE:
            self.index = SparseIndex()
            self.index.data = texts
            built = self._build_embedding_csr(texts, batch_size=batch_size) if len(texts) else None
            if built is not None:
                self.index.adopt_device_index(built, dtype=torch.float32)
            else:
                self.index.vector = self._build_embedding_csr_host(texts, batch_size=batch_size)
        elif index_type == IndexType.BAG_OF_TOKEN:
            self.index = BoTIndex()
            self.index.data = texts
            self.index.vector = self._build_bot_vectors(texts, batch_size=batch_size)
        else:
            raise NotImplementedError
        self.index.move_to_device(self.device)
        if devices is not None:
            if index_type == IndexType.DENSE:
                raise NotImplementedError("devices=: row sharding serves the sparse and bag-of-token indexes")
            self.index.shard_rows(devices)

    def save_index(self, path):
        self.index.save(path)

    def load_index(self, index_file=None, data_file=None, index_type=None, devices=None):
        """retriever.py:322-348.  `devices` (not in the reference): row-shard a sparse / bag-of-token index over several GPUs of this
        process -- None (one device, the reference's behaviour), "all", a count, or a list of GPU ordinals (SparseIndex docstring)."""
        if index_type is None:
            if index_file.endswith(".pt"):
                index_type = IndexType.DENSE
            elif index_file.endswith(".npz"):
                index_type = IndexType.SPARSE
            else:
                raise ValueError("Cannot infer index type from file extension. Please provide 'index_type' explicitly.")
        elif isinstance(index_type, str):
            index_type = IndexType(index_type.lower())
        elif not isinstance(index_type, IndexType):
            raise TypeError("index_type must be an instance of IndexType, int, or str.")
        self.index_type = index_type
        cls = {IndexType.DENSE: Index, IndexType.SPARSE: SparseIndex, IndexType.BAG_OF_TOKEN: BoTIndex}[index_type]
        if devices is not None and index_type == IndexType.DENSE:
            raise NotImplementedError("devices=: row sharding serves the sparse and bag-of-token indexes")
        self.index = cls(index_file, data_file, device=self.device) if devices is None else cls(index_file, data_file, device=self.device, devices=devices)
