"""Thin object wrapper over the ``vs_index`` handle of libvsearch_hip.so.

numpy arrays are passed as host pointers, torch CUDA tensors as device pointers (the library
detects which); results come back in the same kind as the query (numpy in -> numpy out, torch
in -> torch tensors on the index device, on torch's current stream).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat

_NP2VS = {np.dtype(np.float32): nat.VS_F32, np.dtype(np.float16): nat.VS_F16,
          np.dtype(np.int32): nat.VS_I32, np.dtype(np.int64): nat.VS_I64}


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _torch_vs_dtype(t):
    import torch
    return {torch.float32: nat.VS_F32, torch.float16: nat.VS_F16, torch.int32: nat.VS_I32, torch.int64: nat.VS_I64}[t.dtype]


def as_arg(x, allowed=None):
    """-> (pointer, vs dtype, keep-alive object). Accepts numpy arrays and torch tensors (CPU or CUDA)."""
    if x is None:
        return None, nat.VS_NONE, None
    if _is_torch(x):
        t = x.detach()
        if not t.is_contiguous():
            t = t.contiguous()
        dt = _torch_vs_dtype(t)
        if allowed and dt not in allowed:
            raise TypeError(f"unsupported dtype {t.dtype}")
        return C.c_void_p(t.data_ptr()), dt, t
    a = np.ascontiguousarray(x)
    if a.dtype not in _NP2VS:
        raise TypeError(f"unsupported dtype {a.dtype}")
    dt = _NP2VS[a.dtype]
    if allowed and dt not in allowed:
        raise TypeError(f"unsupported dtype {a.dtype}")
    return C.c_void_p(a.ctypes.data), dt, a


def current_stream(device: int):
    """hipStream_t of torch's current stream on `device` (None when torch has no CUDA/HIP device)."""
    try:
        import torch
        if torch.cuda.is_available():
            # the null stream is passed as hipStreamLegacy ((hipStream_t)1): same stream, but not "no stream given" (which blocks)
            return C.c_void_p(torch.cuda.current_stream(device).cuda_stream or 1)
    except Exception:
        pass
    return None


def npz_inspect(path: str, shift: int = 0):
    """(n_rows, n_cols after the shift, nnz after the shift, 8-nnz packets) of a scipy.sparse.save_npz CSR shard; no GPU needed."""
    n_rows, n_cols, nnz, packets = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
    nat.check(nat.lib().vs_npz_inspect(str(path).encode(), int(shift), C.byref(n_rows), C.byref(n_cols), C.byref(nnz), C.byref(packets)))
    return n_rows.value, n_cols.value, nnz.value, packets.value


class DeviceIndex:
    """Owner of one device-resident index shard (CSR packets or dense)."""

    def __init__(self, handle):
        self._h = handle
        self._device = None        # cached: info() reads device-side search statistics and therefore synchronises

    @property
    def device(self) -> int:
        if self._device is None:
            self._device = int(self.info().device)
        return self._device

    # ---- constructors -------------------------------------------------------------------------
    @classmethod
    def from_csr(cls, indptr, indices, data, n_cols, store_dtype=None, device=0):
        """CSR arrays (numpy or torch, host or device). data=None -> binary index.
        store_dtype: VS_F32 | VS_F16 | VS_NONE; default = dtype of `data` (binary when data is None)."""
        nat.require_device()
        p_rp, dt_rp, k1 = as_arg(indptr, (nat.VS_I32, nat.VS_I64))
        p_ci, dt_ci, k2 = as_arg(indices, (nat.VS_I32, nat.VS_I64))
        p_v, dt_v, k3 = as_arg(data, (nat.VS_F32, nat.VS_F16))
        n_rows = int(indptr.shape[0]) - 1
        if store_dtype is None:
            store_dtype = dt_v if data is not None else nat.VS_NONE
        h = C.c_void_p()
        nat.check(nat.lib().vs_index_create_csr(p_rp, dt_rp, p_ci, dt_ci, p_v, dt_v if data is not None else nat.VS_F32,
                                                store_dtype, n_rows, int(n_cols), int(device), C.byref(h)))
        del k1, k2, k3
        return cls(h)

    @classmethod
    def reserved(cls, rows_cap, packets_cap, n_cols, store_dtype, device=0):
        """Empty CSR index with room for rows_cap rows / packets_cap 8-nnz packets; fill with append_csr()."""
        nat.require_device()
        h = C.c_void_p()
        nat.check(nat.lib().vs_index_create_reserved(int(rows_cap), int(packets_cap), int(n_cols), int(store_dtype), int(device), C.byref(h)))
        return cls(h)

    def append_csr(self, indptr, indices, data):
        """Append a block of CSR rows (a shard) behind the rows already in the index."""
        p_rp, dt_rp, k1 = as_arg(indptr, (nat.VS_I32, nat.VS_I64))
        p_ci, dt_ci, k2 = as_arg(indices, (nat.VS_I32, nat.VS_I64))
        p_v, dt_v, k3 = as_arg(data, (nat.VS_F32, nat.VS_F16))
        nat.check(nat.lib().vs_index_append_csr(self._h, p_rp, dt_rp, p_ci, dt_ci, p_v, dt_v if data is not None else nat.VS_F32,
                                                int(indptr.shape[0]) - 1))

    def slice_rows(self, row0: int, n_rows: int, device: int = None) -> "DeviceIndex":
        """Rows [row0, row0 + n_rows) as a new CSR index on GPU `device` (default: this index's): a device-to-device (peer) copy of
        the packets -- what row-range sharding deals out (vs_index_slice_rows)."""
        h = C.c_void_p()
        nat.check(nat.lib().vs_index_slice_rows(self._h, int(row0), int(n_rows), int(self.device if device is None else device), C.byref(h)))
        return DeviceIndex(h)

    def append_npz(self, path: str, shift: int = 0):
        """Append a scipy.sparse.save_npz CSR shard read natively (zip + npy parsed in the library): columns below `shift` dropped,
        ids moved down by `shift`, sorted within a row.  NotImplementedError for non-CSR files; NotBinaryError when a binary
        (bag-of-token) index meets a value other than 1."""
        try:
            nat.check(nat.lib().vs_index_append_npz(self._h, str(path).encode(), int(shift)))
        except ValueError as e:
            if "expects a binary matrix" in str(e):
                raise NotBinaryError(str(e)) from None
            raise

    def save_npz(self, path: str, compressed: bool = False):
        """Write the index as a scipy.sparse.save_npz file (CSR, int64 ids, fp32 data) without scipy."""
        nat.check(nat.lib().vs_index_save_npz(self._h, str(path).encode(), 1 if compressed else 0))

    def save_native(self, path: str):
        """Write the device format verbatim (.vsx shard file)."""
        nat.check(nat.lib().vs_index_save_native(self._h, str(path).encode()))

    @classmethod
    def load_native(cls, path: str, device=0):
        nat.require_device()
        h = C.c_void_p()
        nat.check(nat.lib().vs_index_load_native(str(path).encode(), int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def from_dense(cls, mat, store_dtype=None, device=0, max_density=0.0):
        """Dense [N, V] index. max_density > 0: store as CSR packets when the matrix is that sparse
        (sparsity-aware dense index: same results, searched by the CSR scan)."""
        nat.require_device()
        p, dt, keep = as_arg(mat, (nat.VS_F32, nat.VS_F16))
        n_rows, n_cols = int(mat.shape[0]), int(mat.shape[1])
        h = C.c_void_p()
        nat.check(nat.lib().vs_index_create_dense_auto(p, dt, dt if store_dtype is None else store_dtype, n_rows, n_cols, n_cols,
                                                       float(max_density), int(device), C.byref(h)))
        del keep
        return cls(h)

    @classmethod
    def synthetic(cls, seed, row0, n_rows, n_cols=29523, nnz=768, kind=0, val_law=0, store_dtype=nat.VS_F32, device=0):
        nat.require_device()
        h = C.c_void_p()
        nat.check(nat.lib().vs_index_create_synthetic(C.c_uint64(seed), int(row0), int(n_rows), int(n_cols), int(nnz), int(kind),
                                                      int(val_law), int(store_dtype), int(device), C.byref(h)))
        return cls(h)

    # ---- lifetime -------------------------------------------------------------------------------
    def prepare(self):
        """Build now what the first sparse search would build inside the call (vs_index_prepare)."""
        nat.check(nat.lib().vs_index_prepare(self._h, None))
        return self

    def close(self):
        if self._h is not None and self._h.value:
            nat.lib().vs_index_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- queries --------------------------------------------------------------------------------
    def info(self) -> nat.IndexInfo:
        out = nat.IndexInfo()
        nat.check(nat.lib().vs_index_info(self._h, C.byref(out)))
        return out

    def set_queries_per_pass(self, qt: int):
        """0 = auto (tiles of 8 sparse queries per index pass when the batch qualifies), 1 = one query per pass."""
        nat.check(nat.lib().vs_index_set_queries_per_pass(self._h, int(qt)))

    def set_option(self, name: str, value: int):
        """Tuning / test options: "blocked_postings" (-1 auto, 0 off, 1 on), "mq_variant" (-1 auto, 0 plain, 1 shared columns)."""
        nat.check(nat.lib().vs_index_set_option(self._h, name.encode(), int(value)))

    def _q_args(self, q):
        if q.ndim != 2:
            raise ValueError("queries must be [B, V]")
        p, dt, keep = as_arg(q, (nat.VS_F32, nat.VS_F16))
        return p, dt, keep, int(q.shape[0]), int(q.shape[1])

    def search(self, q, k: int, id_offset: int = 0):
        """Top-k per query -> (ids int64 [B,k], scores float32 [B,k]); canonical order (score desc, id asc).
        Device queries: the call returns once the kernels are enqueued on torch's current stream (no host synchronisation on
        the postings filter path); host queries / outputs are copied and the call blocks."""
        p, dt, keep, B, ldq = self._q_args(q)
        k = int(k)
        if _is_torch(q) and q.is_cuda:
            import torch
            dev = torch.device("cuda", self.device)
            ids = torch.empty((B, k), dtype=torch.int64, device=dev)
            scores = torch.empty((B, k), dtype=torch.float32, device=dev)
            stream = current_stream(self.device)
            nat.check(nat.lib().vs_index_search(self._h, p, dt, ldq, B, k, int(id_offset), C.c_void_p(ids.data_ptr()),
                                                C.c_void_p(scores.data_ptr()), stream))
            return ids, scores
        ids = np.empty((B, k), dtype=np.int64)
        scores = np.empty((B, k), dtype=np.float32)
        nat.check(nat.lib().vs_index_search(self._h, p, dt, ldq, B, k, int(id_offset), C.c_void_p(ids.ctypes.data),
                                            C.c_void_p(scores.ctypes.data), None))
        if _is_torch(q):
            import torch
            return torch.from_numpy(ids), torch.from_numpy(scores)
        return ids, scores

    def scores(self, q):
        """Dense [B, n_rows] fp32 score matrix (what index.py:91 materialises). numpy out."""
        info = self.info()
        p, dt, keep, B, ldq = self._q_args(q)
        out = np.empty((B, info.n_rows), dtype=np.float32)
        nat.check(nat.lib().vs_index_scores(self._h, p, dt, ldq, B, C.c_void_p(out.ctypes.data), None))
        return out

    def export_csr(self, val_dtype=np.float32):
        """-> (indptr int64, indices int64, data) as numpy arrays (host)."""
        info = self.info()
        indptr = np.empty(info.n_rows + 1, dtype=np.int64)
        nat.check(nat.lib().vs_index_export_csr(self._h, C.c_void_p(indptr.ctypes.data), None, None, nat.VS_F32))
        nnz = int(indptr[-1])
        indices = np.empty(nnz, dtype=np.int64)
        data = np.empty(nnz, dtype=val_dtype)
        nat.check(nat.lib().vs_index_export_csr(self._h, C.c_void_p(indptr.ctypes.data), C.c_void_p(indices.ctypes.data),
                                                C.c_void_p(data.ctypes.data), _NP2VS[np.dtype(val_dtype)]))
        return indptr, indices, data

    def export_dense(self, dtype=np.float32):
        info = self.info()
        out = np.empty((info.n_rows, info.n_cols), dtype=dtype)
        nat.check(nat.lib().vs_index_export_dense(self._h, C.c_void_p(out.ctypes.data), _NP2VS[np.dtype(dtype)], info.n_cols))
        return out


class NotBinaryError(ValueError):
    """a binary (bag-of-token) index was given a stored value other than 1"""


class ShardGroup:
    """Row-sharded search inside one process (``vs_shard_group_*``): one DeviceIndex per GPU holding consecutive row ranges of one
    corpus; search() scores the batch on every GPU, gathers B * k pairs per shard on the first shard's GPU and merges there."""

    def __init__(self, shards):
        nat.require_device()
        self._shards = list(shards)                      # keep them alive: the group does not own the handles
        arr = (C.c_void_p * len(self._shards))(*[s._h for s in self._shards])
        h = C.c_void_p()
        nat.check(nat.lib().vs_shard_group_create(arr, len(self._shards), C.byref(h)))
        self._h = h

    def search(self, q, k: int):
        if q.ndim != 2:
            raise ValueError("queries must be [B, V]")
        p, dt, keep = as_arg(q, (nat.VS_F32, nat.VS_F16))
        B, ldq, k = int(q.shape[0]), int(q.shape[1]), int(k)
        if _is_torch(q) and q.is_cuda:
            import torch
            dev = torch.device("cuda", self._shards[0].device)
            ids = torch.empty((B, k), dtype=torch.int64, device=dev)
            sc = torch.empty((B, k), dtype=torch.float32, device=dev)
            nat.check(nat.lib().vs_shard_group_search(self._h, p, dt, ldq, B, k, C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr())))
            return ids, sc
        ids = np.empty((B, k), dtype=np.int64)
        sc = np.empty((B, k), dtype=np.float32)
        nat.check(nat.lib().vs_shard_group_search(self._h, p, dt, ldq, B, k, C.c_void_p(ids.ctypes.data), C.c_void_p(sc.ctypes.data)))
        return ids, sc

    def close(self):
        if self._h:
            nat.lib().vs_shard_group_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk(cand_ids, cand_scores, k: int, device: int = 0):
    """Canonical top-k of gathered per-shard candidates [B, n_cand] (global ids + scores)."""
    nat.require_device()
    p_i, _, k1 = as_arg(cand_ids, (nat.VS_I64,))
    p_s, _, k2 = as_arg(cand_scores, (nat.VS_F32,))
    B, n = int(cand_ids.shape[0]), int(cand_ids.shape[1])
    if _is_torch(cand_ids) and cand_ids.is_cuda:
        import torch
        ids = torch.empty((B, k), dtype=torch.int64, device=cand_ids.device)
        sc = torch.empty((B, k), dtype=torch.float32, device=cand_ids.device)
        nat.check(nat.lib().vs_merge_topk(p_i, p_s, B, n, int(k), C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()),
                                          int(device), current_stream(device)))
        return ids, sc
    ids = np.empty((B, k), dtype=np.int64)
    sc = np.empty((B, k), dtype=np.float32)
    nat.check(nat.lib().vs_merge_topk(p_i, p_s, B, n, int(k), C.c_void_p(ids.ctypes.data), C.c_void_p(sc.ctypes.data), int(device), None))
    if _is_torch(cand_ids):
        import torch
        return torch.from_numpy(ids), torch.from_numpy(sc)
    return ids, sc


class Profile:
    """bench.py hook: hipEvent timing of the scoring kernels (vs_profile_*)."""

    @staticmethod
    def enable(on=True):
        nat.check(nat.lib().vs_profile_enable(int(bool(on))))

    @staticmethod
    def reset():
        nat.check(nat.lib().vs_profile_reset())

    @staticmethod
    def read(kernel: str):
        ms, n = C.c_double(0), C.c_int64(0)
        nat.check(nat.lib().vs_profile_read(kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
