"""Index containers with the reference's interface (/root/reference/src/ir/retriever/index.py:16-218),
backed by the device-resident formats of libvsearch_hip.so.

Same names, constructor arguments and return conventions as ``src.ir.retriever.index``:
``Index`` (dense), ``SparseIndex`` (CSR), ``BoTIndex`` (binary bag-of-token CSR), ``SearchResults``,
``IndexType``.  Differences, all supersets:
  * ``search`` never materialises the [B, N] score matrix; ties are ordered (score desc, id asc);
  * compute always runs on an MI355X: an index whose ``device`` is "cpu" keeps its API tensors on
    the host but is searched on GPU 0 (there is no CPU fallback -- without a GPU it raises);
  * loading a dense index from ``.pt`` shards and ``low_memory=True`` work (broken upstream);
  * ``fp16=True`` is applied on the device (scipy >= 1.15 cannot ``astype(float16)`` a sparse array).
"""
from __future__ import annotations

import glob
import json
import logging
from enum import Enum
from typing import List, NamedTuple, Optional

import numpy as np
import torch

from ... import _native as nat
from ...device_index import DeviceIndex, NotBinaryError, ShardGroup

logger = logging.getLogger(__name__)


class SearchResults(NamedTuple):
    ids: List[int]
    scores: List[float]


class IndexType(Enum):
    DENSE = "dense"
    SPARSE = "sparse"
    BAG_OF_TOKEN = "bag_of_token"


def _resolve_devices(devices):
    """`devices` of SparseIndex / BoTIndex / Retriever.load_index -> list of GPU ordinals to shard over, or None (one device).
    None: the index lives on `device` (the reference's behaviour); "all": every visible GPU; an int n: GPUs 0 .. n-1; a list."""
    if devices is None:
        return None
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError('devices: None, "all", a count or a list of GPU ordinals')
        devices = torch.cuda.device_count()
    if isinstance(devices, int):
        devices = list(range(devices))
    out = [_gpu_ordinal(d) if not isinstance(d, int) else int(d) for d in devices]
    return out if len(out) > 1 else None


def _gpu_ordinal(device) -> int:
    d = torch.device(device) if not isinstance(device, torch.device) else device
    if d.type == "cuda":
        return d.index if d.index is not None else torch.cuda.current_device()
    return 0


class Index:
    index_type = IndexType.DENSE
    DENSE_AS_CSR_DENSITY = 0.05        # set to 0 to always use the dense (MFMA) search kernel

    def __init__(self, index_file: Optional[str] = None, data_file: Optional[str] = None, fp16: bool = True,
                 device: str = "cpu", low_memory: bool = False):
        self.data = None
        self.low_memory = low_memory
        self.device = device
        self._vector = None            # torch view of the index (dense tensor / sparse CSR tensor), lazily exported
        self._dev: Optional[DeviceIndex] = None
        self._dtype = torch.float32    # dtype the reference would report for `vector` (scores are returned in it)
        self._shape = None
        self.init_index(index_file, fp16)
        self.load_data(data_file)

    # ---- the `vector` attribute of the reference: assignable and readable ----------------------
    @property
    def vector(self):
        if self._vector is None and self._dev is not None:
            self._vector = self._export_vector()
        return self._vector

    @vector.setter
    def vector(self, value):
        self._drop_device()
        if value is not None and not isinstance(value, torch.Tensor):      # scipy CSR
            value = torch.sparse_csr_tensor(torch.from_numpy(value.indptr), torch.from_numpy(value.indices),
                                            torch.from_numpy(value.data), size=value.shape)
        self._vector = value
        if value is not None:
            self._dtype = value.dtype
            self._shape = tuple(value.shape)

    def _drop_device(self):
        if self._dev is not None:
            self._dev.close()
        self._dev = None

    def adopt_device_index(self, dev: DeviceIndex, dtype=torch.float32, device: Optional[str] = None):
        """Take ownership of an index already built in HBM (Retriever.build_index emits the CSR per batch straight into the
        device format): `vector` is exported from it on demand."""
        self._drop_device()
        info = dev.info()
        self._vector = None
        self._dev = dev
        self._dtype = dtype
        self._shape = (int(info.n_rows), int(info.n_cols))
        self.device = device or f"cuda:{info.device}"
        self._prepare()

    def _export_vector(self):
        mat = self._dev.export_dense(np.float16 if self._dtype == torch.float16 else np.float32)
        return torch.from_numpy(mat).to(self.device)

    def _build_device_index(self) -> DeviceIndex:
        v = self._vector
        if v is None:
            raise RuntimeError("index is empty: no vector to search")
        if v.layout != torch.strided:
            raise TypeError(f"{type(self).__name__} expects a dense tensor, got layout {v.layout}")
        if v.dtype not in (torch.float32, torch.float16):
            v = v.float()
        store = nat.VS_F16 if v.dtype == torch.float16 else nat.VS_F32
        # a dense index of VDR embeddings is > 97 % zeros: below DENSE_AS_CSR_DENSITY it is stored as CSR packets
        return DeviceIndex.from_dense(v.contiguous(), store_dtype=store, device=_gpu_ordinal(self.device),
                                      max_density=self.DENSE_AS_CSR_DENSITY)

    # The column-grouped copy sparse queries are searched on is built when the index reaches the device (move_to_device, load_index,
    # build_index) -- 0.5 s at 21 M docs that would otherwise sit inside a user's first retrieve().
    EAGER_POSTINGS = True

    def _prepare(self):
        if self._dev is not None and self.EAGER_POSTINGS:
            self._dev.prepare()

    def _device_index(self) -> DeviceIndex:
        if self._dev is None:
            self._dev = self._build_device_index()
            self._prepare()
        return self._dev

    # ---- loading -------------------------------------------------------------------------------
    def init_index(self, index_path: Optional[str], fp16: bool = True):
        if not index_path:
            return
        files = sorted(glob.glob(index_path))
        if not files:
            raise FileNotFoundError(f"no index file matches {index_path!r}")
        logger.info("***** Loading %s Index from %d files *****", self.index_type.value, len(files))
        shards = [torch.load(f, map_location="cpu") for f in files]
        vector = torch.cat(shards, dim=0) if len(shards) > 1 else shards[0]
        self.vector = vector.to(torch.float16) if fp16 else vector
        self.move_to_device(self.device)

    def load_data(self, data_file: Optional[str]):
        if not data_file:
            return
        if not self.low_memory:
            with open(data_file, "r") as fh:
                self.data = [json.loads(line) for line in fh]
        else:
            self.data_file = data_file
            self.offsets = self._calculate_offsets(data_file)

    def move_to_device(self, device: str):
        logger.info("Moving index to %s.", device)
        if self._vector is not None:
            self._vector = self._vector.to(device)
        same_gpu = self._dev is not None and self._dev.device == _gpu_ordinal(device)
        if not same_gpu:
            if self._vector is None and self._dev is not None:
                self._vector = self._export_vector()
            self._drop_device()
        self.device = device
        if self._vector is not None or self._dev is not None:
            self._device_index()
            if torch.device(device).type == "cuda" and self._vector is not None and self._keep_only_device_copy():
                self._vector = None            # re-exported on demand; avoids a second device copy

    def _keep_only_device_copy(self) -> bool:
        return True

    # ---- text store ----------------------------------------------------------------------------
    @staticmethod
    def _calculate_offsets(data_file: str):
        offsets, pos = [], 0
        with open(data_file, "rb") as fh:
            for line in fh:
                offsets.append(pos)
                pos += len(line)
        return offsets

    def _load_line(self, file_path: str, offset: int):
        with open(file_path, "rb") as fh:
            fh.seek(offset)
            return json.loads(fh.readline().decode("utf-8"))

    def get_sample(self, index: int):
        if not self.low_memory:
            return self.data[index]
        return self._load_line(self.data_file, self.offsets[index])

    # ---- search (index.py:88-94) -------------------------------------------------------------------
    def search(self, q_embs: torch.Tensor, k: int) -> SearchResults:
        if isinstance(q_embs, np.ndarray):
            q_embs = torch.from_numpy(q_embs)
        dev_index = self._device_index()
        gpu = torch.device("cuda", dev_index.device)         # (cached: info() reads search statistics and synchronises)
        q = q_embs.detach().to(gpu)
        q = q.to(self._dtype) if self._dtype in (torch.float16, torch.float32) else q.float()   # `.type(self.vector.dtype)`
        if q.dim() == 1:
            q = q.unsqueeze(0)
        ids, scores = dev_index.search(q.contiguous(), int(k))
        scores = scores.to(self._dtype)
        if torch.device(self.device).type != "cuda":
            ids, scores = ids.cpu(), scores.cpu()
        return SearchResults(ids, scores)

    # ---- persistence -------------------------------------------------------------------------------
    def save(self, path):
        """Dense index -> ``.pt`` (torch.save of the CPU tensor), like index.py:96-109."""
        try:
            torch.save(self.vector.cpu(), path)
            logger.info("Index successfully saved to %s", path)
        except Exception as exc:
            logger.error("Failed to save index to %s: %s", path, exc)
            raise

    def __len__(self):
        if self.data:
            return len(self.data)
        return len(getattr(self, "offsets", [])) if self.low_memory else 0

    def __repr__(self):
        return repr(self.vector)

    def _vector_meta(self):
        if self._vector is not None:
            return self._vector.shape, self._vector.dtype, self._vector.layout
        if self._dev is not None:
            info = self._dev.info()
            layout = torch.strided if info.kind == nat.VS_KIND_DENSE else torch.sparse_csr
            return torch.Size([info.n_rows, info.n_cols]), self._dtype, layout
        if getattr(self, "_shards", None):
            return torch.Size(self._shape), self._dtype, torch.sparse_csr
        return None, None, None

    def __str__(self):
        shape, dtype, layout = self._vector_meta()
        rows = [("Index Type", type(self).__name__), ("Vector Shape", shape), ("Vector Dtype", dtype),
                ("Vector Layout", layout), ("Number of Texts", len(self.data) if self.data else 0), ("Device", self.device)]
        return "".join(f"{name:<18}: {value}\n" for name, value in rows)


def _vsx_rows(path: str) -> int:
    """rows of a native .vsx shard, from its header (csr_index.hip VsxHeader: magic[8], int32 store_dtype, n_cols, int64 n_rows, ...)"""
    import struct
    with open(path, "rb") as fh:
        hdr = fh.read(24)
    if len(hdr) < 24 or hdr[:7] != b"VSXCSR1":
        raise ValueError(f"{path} is not a vsearch native shard file")
    return int(struct.unpack_from("<q", hdr, 16)[0])


class SparseIndex(Index):
    index_type = IndexType.SPARSE

    def __init__(self, index_file: Optional[str] = None, data_file: Optional[str] = None, fp16: bool = True,
                 device: str = "cpu", low_memory: bool = False, shift: int = 0, devices=None):
        """`devices` (not in the reference: its index lives on one device) row-shards the index over several GPUs of this process:
        the shard files matched by `index_file` are dealt to the GPUs in row order (whole files; the reference builds Wiki21M as
        per-shard .npz files, examples/inference_sparse/README.md:90-107, and re-joins them with vstack at index.py:172-175),
        `search` scores the batch on every GPU and merges the per-shard top-k on the first (vs_shard_group_*).  Results are
        identical to the unsharded index."""
        self.shift = shift
        self._devices = _resolve_devices(devices)
        self._shards: Optional[List[DeviceIndex]] = None
        self._group: Optional[ShardGroup] = None
        super().__init__(index_file, data_file, fp16, device, low_memory)

    # -- row shards over several GPUs ------------------------------------------------------------------
    def _drop_device(self):
        if self._group is not None:
            self._group.close()
        for sh in (self._shards or []):
            sh.close()
        self._group, self._shards = None, None
        super()._drop_device()

    def _adopt_shards(self, shards: List[DeviceIndex]):
        self._shards = shards
        self._group = ShardGroup(shards)
        self._dev = None
        if self.EAGER_POSTINGS:
            for sh in shards:
                sh.prepare()
        self.device = f"cuda:{shards[0].device}"

    def shard_rows(self, devices):
        """Deal this (unsharded, device-resident) index over `devices` in contiguous, equal row ranges ("per-shard npz
        or row ranges"): one .npz / .vsx file -- what SparseIndex.save writes, index.py:181-202 -- or an index built in memory
        (Retriever.build_index, retriever.py:284-317) then uses more than one GPU.  Device-to-device copies of the packets
        (vs_index_slice_rows); results stay bit-identical to the unsharded index."""
        gpus = _resolve_devices(devices)
        if not gpus:
            return self
        if self._shards:
            raise NotImplementedError("the index is row-sharded already")
        from vsearch_amd.distributed import shard_rows as _range
        whole = self._device_index()
        info = whole.info()
        if info.kind != nat.VS_KIND_CSR:
            raise NotImplementedError("row sharding serves the sparse and bag-of-token indexes")
        self._shape = (int(info.n_rows), int(info.n_cols))
        built = []
        try:
            for r, gpu in enumerate(gpus):
                row0, n = _range(int(info.n_rows), len(gpus), r)
                built.append(whole.slice_rows(row0, n, gpu))
        except Exception:
            for dev in built:
                dev.close()
            raise
        self._dev = None
        self._vector = None
        whole.close()
        self._devices = gpus
        self._adopt_shards(built)
        return self

    @property
    def shards(self):
        """the DeviceIndex of every row shard, in row order (None: the index is not sharded)"""
        return list(self._shards) if self._shards else None

    @property
    def shard_devices(self):
        """GPU ordinal of every row shard (None: the index is not sharded)"""
        return [sh.device for sh in self._shards] if self._shards else None

    def search(self, q_embs: torch.Tensor, k: int) -> SearchResults:
        if self._group is None:
            return super().search(q_embs, k)
        if isinstance(q_embs, np.ndarray):
            q_embs = torch.from_numpy(q_embs)
        gpu = torch.device("cuda", self._shards[0].device)
        q = q_embs.detach().to(gpu)
        q = q.to(self._dtype) if self._dtype in (torch.float16, torch.float32) else q.float()
        if q.dim() == 1:
            q = q.unsqueeze(0)
        ids, scores = self._group.search(q.contiguous(), int(k))
        return SearchResults(ids, scores.to(self._dtype))

    def move_to_device(self, device: str):
        if self._group is not None:
            if torch.device(device).type == "cuda" and _gpu_ordinal(device) in self.shard_devices:
                return                                              # (already there: the shards stay where they are)
            raise NotImplementedError("a row-sharded index stays on its GPUs; reload it with device= / devices= to move it")
        super().move_to_device(device)

    # -- conversions ---------------------------------------------------------------------------------
    @staticmethod
    def _csr_parts(v):
        """torch sparse CSR tensor | scipy CSR -> (indptr, indices, data, (n_rows, n_cols))."""
        if isinstance(v, torch.Tensor):
            if v.layout != torch.sparse_csr:
                v = v.to_sparse_csr()
            return v.crow_indices(), v.col_indices(), v.values(), tuple(v.shape)
        return v.indptr, v.indices, v.data, tuple(v.shape)         # scipy.sparse.csr_array / csr_matrix

    def _binary(self) -> bool:
        return False

    def _build_device_index(self) -> DeviceIndex:
        v = self._vector
        if v is None:
            raise RuntimeError("index is empty: no vector to search")
        indptr, indices, data, shape = self._csr_parts(v)
        if isinstance(data, torch.Tensor) and data.dtype not in (torch.float32, torch.float16):
            data = data.float()
        if isinstance(data, np.ndarray) and data.dtype not in (np.float32, np.float16):
            data = data.astype(np.float32)
        if self._binary() and bool((data == 1).all()):
            store, data = nat.VS_NONE, None                          # ids only: 2 bytes per non-zero, integer-exact scores
        else:
            # (a BoTIndex over a VALUED matrix: the reference searches it like any sparse index, index.py:205-218 -- so do we)
            store = nat.VS_F16 if self._dtype == torch.float16 else nat.VS_F32
        return DeviceIndex.from_csr(indptr, indices, data, shape[1], store_dtype=store, device=_gpu_ordinal(self.device))

    def _export_vector(self):
        parts = self._shards if self._shards else [self._dev]
        ip_all, ix_all, d_all, rows, nnz0, n_cols = [np.zeros(1, np.int64)], [], [], 0, 0, 0
        for dev in parts:
            indptr, indices, data = dev.export_csr(np.float16 if self._dtype == torch.float16 else np.float32)
            info = dev.info()
            ip_all.append(indptr[1:] + nnz0)
            ix_all.append(indices)
            d_all.append(data)
            rows += info.n_rows
            nnz0 += int(indptr[-1])
            n_cols = info.n_cols
        t = torch.sparse_csr_tensor(torch.from_numpy(np.concatenate(ip_all)), torch.from_numpy(np.concatenate(ix_all)),
                                    torch.from_numpy(np.concatenate(d_all)), size=(rows, n_cols))
        return t.to(self.device)

    @property
    def vector(self):
        if self._vector is None and (self._dev is not None or self._shards):
            self._vector = self._export_vector()
        return self._vector

    @vector.setter
    def vector(self, value):
        Index.vector.fset(self, value)

    def _scipy_csr_to_torch_csr(self, mat) -> torch.Tensor:
        t = torch.sparse_csr_tensor(torch.from_numpy(mat.indptr), torch.from_numpy(mat.indices), torch.from_numpy(mat.data), size=mat.shape)
        return t.to(self.device)

    # -- loading (index.py:163-179) ------------------------------------------------------------------
    def init_index(self, index_file: Optional[str], fp16: bool = True):
        if not index_file:
            return
        from scipy.sparse import load_npz
        files = sorted(glob.glob(index_file))
        if not files:
            raise FileNotFoundError(f"no index file matches {index_file!r}")
        logger.info("***** Loading %s Index from %d files *****", self.index_type.value, len(files))
        if all(f.endswith(".vsx") for f in files) and self._devices and len(files) < len(self._devices):
            # fewer native files than GPUs: row ranges.  The files are joined on the first GPU (device-to-device), then dealt out
            if self.shift:
                raise ValueError(f"a native .vsx shard stores the columns after the shift was applied: load it with shift=0 (got shift={self.shift})")
            self._drop_device()
            self._vector = None
            gpus, self._devices = self._devices, None
            try:
                if len(files) == 1:
                    self.init_index(files[0], fp16)
                    self.shard_rows(gpus)
                else:
                    # every file's rows, sliced at the GPUs' range boundaries; a GPU whose range spans two files holds two shards
                    # (a shard group takes any number of shards per device, in row order).  ONE file at a time (all files at
                    # once on the first GPU needed room for the whole index there): the row counts come from the files' headers, a file
                    # is loaded on the GPU that gets most of its rows, cut into the ranges it overlaps (peer copies) and closed
                    from vsearch_amd.distributed import shard_rows as _range
                    rows = [_vsx_rows(f) for f in files]
                    total, built, start = sum(rows), [], 0
                    bounds = [_range(total, len(gpus), r) for r in range(len(gpus))]
                    try:
                        for f, n in zip(files, rows):
                            cuts = [(gpu, max(start, b0), min(start + n, b0 + bn)) for gpu, (b0, bn) in zip(gpus, bounds)]
                            cuts = [(gpu, lo, hi) for gpu, lo, hi in cuts if hi > lo]
                            home = max(cuts, key=lambda c: c[2] - c[1])[0] if cuts else gpus[0]
                            dev = DeviceIndex.load_native(f, device=home)
                            try:
                                if int(dev.info().n_rows) != n:
                                    raise ValueError(f"{f}: header says {n} rows, the file holds {int(dev.info().n_rows)}")
                                for gpu, lo, hi in cuts:
                                    built.append(dev.slice_rows(lo - start, hi - lo, gpu))
                            finally:
                                dev.close()
                            start += n
                    except Exception:
                        for d in built:
                            d.close()
                        raise
                    info = built[0].info()
                    self._dtype = torch.float32 if info.store_dtype == nat.VS_F32 else torch.float16
                    self._shape = (total, int(info.n_cols))
                    self._adopt_shards(built)
            finally:
                self._devices = gpus
            return
        if all(f.endswith(".vsx") for f in files) and (len(files) == 1 or self._devices):
            # native shard files: the device format verbatim.  One file -> this device; several (with `devices`) -> one row shard each,
            # dealt to the GPUs in order
            if self.shift:
                raise ValueError("a native .vsx shard stores the columns after the shift was applied: load it with shift=0 "
                                 f"(got shift={self.shift}); convert from .npz shards to change it")
            self._drop_device()
            self._vector = None
            gpus = self._devices or [_gpu_ordinal(self.device)]
            loaded = []
            try:
                for i, f in enumerate(files):
                    dev = DeviceIndex.load_native(f, device=gpus[i * len(gpus) // len(files)])
                    loaded.append(dev)
                    info = dev.info()
                    # the file holds the device format verbatim: it must be the kind of index this class searches
                    if info.kind != nat.VS_KIND_CSR:
                        raise ValueError(f"{f} holds a dense index (sparsity-aware dense store): load it with Index, not {type(self).__name__}")
                    if not self._binary() and info.store_dtype == nat.VS_NONE:   # (a BoTIndex may hold a valued matrix, like the reference's)
                        raise ValueError(f"{f} holds a binary (bag-of-token) index: it cannot be loaded as {type(self).__name__}")
                    if loaded[0].info().store_dtype != info.store_dtype or loaded[0].info().n_cols != info.n_cols:
                        raise ValueError(f"{f}: shards of one index must agree in value type and column count")
            except Exception:
                for dev in loaded:
                    dev.close()
                raise
            info = loaded[0].info()
            if not fp16 and info.store_dtype == nat.VS_F16:
                logger.warning("%s stores fp16 values; fp16=False cannot restore fp32 precision", files[0])
            self._dtype = torch.float32 if info.store_dtype == nat.VS_F32 else torch.float16
            self._shape = (sum(d.info().n_rows for d in loaded), info.n_cols)
            if len(loaded) == 1:
                self._dev = loaded[0]
                self._prepare()                                     # (after the checks: a rejected file costs no second copy in HBM)
            else:
                self._adopt_shards(loaded)
            return
        # pass 1: sizes only.  CSR shards are inspected by the library (zip + npy parsed natively, the column shift applied);
        # other scipy formats go through scipy (bound by nnz).
        from vsearch_amd.device_index import npz_inspect
        rows_total, packets_cap, n_cols = 0, 0, None
        native, f_rows, f_packets = {}, {}, {}
        for f in files:
            try:
                n_r, n_c, _, pk = npz_inspect(f, self.shift)
                native[f] = True
                shape = (n_r, n_c + self.shift)
            except NotImplementedError:                             # not a CSR file
                native[f] = False
                with np.load(f) as z:
                    shape = tuple(int(x) for x in z["shape"])
                    pk = int(z["data"].shape[0]) if "data" in z.files else shape[0] * shape[1]
            f_rows[f], f_packets[f] = shape[0], pk
            packets_cap += pk
            rows_total += shape[0]
            if n_cols is None:
                n_cols = shape[1]
            elif n_cols != shape[1]:
                raise ValueError(f"shard {f} has {shape[1]} columns, expected {n_cols}")
        self._drop_device()
        self._vector = None
        self._dtype = torch.float16 if fp16 else torch.float32
        self._shape = (rows_total, n_cols - self.shift)
        logger.info("***** Converting Sparse index to the device CSR format *****")

        # pass 2: one shard at a time -- the reference's vstack(shards) (index.py:175) never exists on the host.  With `devices` the
        # files are dealt to the GPUs in row order, whole files, as evenly in rows as their boundaries allow.
        gpus = self._devices or [_gpu_ordinal(self.device)]
        split_rows = bool(self._devices) and len(files) < len(gpus)      # fewer files than GPUs (one .npz: what `save` writes): row ranges
        all_gpus = gpus
        if split_rows and len(files) == 1:
            gpus = gpus[:1]
        groups, start, g_row0 = [[] for _ in gpus], 0, [None] * len(gpus)
        for f in files:
            # (row ranges: a file goes to the GPU whose range holds its middle row -- converted there, then cut into the ranges it overlaps;
            #  whole files: to the GPUs in row order, as evenly in rows as the files' boundaries allow)
            at = (start + f_rows[f] // 2) if split_rows else start
            g = min(len(gpus) - 1, at * len(gpus) // max(rows_total, 1))
            if g_row0[g] is None:
                g_row0[g] = start
            groups[g].append(f)
            start += f_rows[f]
        plan = [(g, fs) for g, fs in zip(gpus, groups) if fs]
        plan_row0 = [r0 for r0, fs in zip(g_row0, groups) if fs]

        def convert(store):
            built = []
            try:
                for gpu, fs in plan:
                    dev = DeviceIndex.reserved(sum(f_rows[f] for f in fs), sum(f_packets[f] for f in fs), n_cols - self.shift, store, device=gpu)
                    built.append(dev)
                    for f in fs:
                        if native[f]:
                            dev.append_npz(f, self.shift)           # file -> rows, no scipy object in between
                            continue
                        mat = load_npz(f).tocsr()[:, self.shift:]
                        mat.sort_indices()
                        data = mat.data.astype(np.float32, copy=False)
                        if store == nat.VS_NONE:
                            if not bool((data == 1).all()):
                                raise NotBinaryError("BoTIndex expects a binary matrix (every stored value == 1)")
                            data = None
                        dev.append_csr(mat.indptr, mat.indices, data)
            except Exception:
                for dev in built:                                   # (nothing reserved survives a failed conversion)
                    dev.close()
                raise
            return built

        valued = nat.VS_F16 if fp16 else nat.VS_F32                  # fp32 -> fp16 happens on the device
        if self._binary():
            try:
                built = convert(nat.VS_NONE)
            except NotBinaryError:
                built = None
            if built is None:
                # shards with values other than 1: the reference's BoTIndex searches them like any sparse index (index.py:205-218)
                logger.info("%s: the shards hold values other than 1 -- stored as a valued sparse index", type(self).__name__)
                built = convert(valued)
        else:
            built = convert(valued)
        if len(built) == 1 and split_rows:
            self._dev = built[0]
            self.shard_rows(all_gpus)                                # joined on the first GPU, dealt out in row ranges (device to device)
        elif split_rows:
            # several files, fewer than GPUs: every converted group is cut into the GPUs' row ranges (peer copies) and closed; when a GPU has
            # no room for a range beside the group it holds, the groups stay as they are -- whole files as shards (the dealing before round 5)
            from vsearch_amd.distributed import shard_rows as _range
            bounds = [_range(rows_total, len(all_gpus), r) for r in range(len(all_gpus))]
            cut, ok = [], True
            try:
                for dev, r0 in zip(built, plan_row0):
                    n = int(dev.info().n_rows)
                    for gpu, (b0, bn) in zip(all_gpus, bounds):
                        lo, hi = max(r0, b0), min(r0 + n, b0 + bn)
                        if hi > lo:
                            cut.append(dev.slice_rows(lo - r0, hi - lo, gpu))
            except nat.VsearchNativeError as e:
                logger.warning("row ranges over %d GPUs did not fit (%s): keeping %d whole-file shards", len(all_gpus), e, len(built))
                ok = False
                for d in cut:
                    d.close()
            if ok:
                for dev in built:
                    dev.close()
                built = cut
            self._adopt_shards(built)
        elif len(built) == 1:
            self._dev = built[0]
            self._prepare()
        else:
            self._adopt_shards(built)

    # -- persistence (index.py:181-202) --------------------------------------------------------------
    def save(self, path):
        """CSR index -> scipy ``.npz`` (keys indices, indptr, data, shape, format; int64 indices)."""
        from scipy.sparse import csr_array, save_npz
        if str(path).endswith(".vsx"):                              # native shard file (loads without a CSR round trip)
            if self._shards:
                raise NotImplementedError("a row-sharded index is saved shard by shard: index.shards[i].save_native(path_i)")
            self._device_index().save_native(path)
            logger.info("Index successfully saved to %s", path)
            return
        try:
            # values go to disk as float32: scipy.sparse has no float16, and the loader re-applies fp16 (fp16=True)
            if self._dev is not None and not self._shards:
                # written by the library (csrc/npz.hip): the same keys / dtypes scipy.sparse.save_npz writes for the reference's
                # int64 torch CSR, stored (uncompressed) members; numpy's savez appends ".npz" to a bare name, so does this
                target = str(path) if str(path).endswith(".npz") else str(path) + ".npz"
                self._dev.save_npz(target, compressed=False)
                logger.info("Index successfully saved to %s", target)
                return
            else:
                ip, ix, d, shape = self._csr_parts(self.vector)        # (a row-sharded index: its shards' rows re-joined)
                indptr, indices, data = ip.cpu().numpy(), ix.cpu().numpy(), d.float().cpu().numpy()
            save_npz(path, csr_array((data, indices, indptr), shape=shape))
            logger.info("Index successfully saved to %s", path)
        except Exception as exc:
            logger.error("Failed to save index to %s: %s", path, exc)
            raise


class BoTIndex(SparseIndex):
    index_type = IndexType.BAG_OF_TOKEN

    def __init__(self, index_file: Optional[str] = None, data_file: Optional[str] = None, fp16: bool = True,
                 device: str = "cpu", low_memory: bool = False, shift: int = 0, devices=None):
        super().__init__(index_file, data_file, fp16, device, low_memory, shift, devices)

    def _binary(self) -> bool:
        return True
