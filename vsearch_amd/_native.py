"""ctypes binding of ``libvsearch_hip.so`` (C ABI: ``include/vsearch_hip.h``).

The product path has no CPU fallback: if the shared library is missing, or no HIP device is
visible, every compute entry point raises -- loudly -- instead of routing elsewhere.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvsearch_hip.so")
CSRC = os.path.join(_HERE, "csrc")

# error codes / dtypes (mirror include/vsearch_hip.h)
VS_OK, VS_EINVAL, VS_ERANGE, VS_ENOMEM, VS_EHIP, VS_EUNSUPPORTED, VS_ENODEVICE = 0, -1, -2, -3, -4, -5, -6
VS_F32, VS_F16, VS_I32, VS_I64, VS_U16, VS_U8, VS_NONE = 0, 1, 2, 3, 4, 5, -1
VS_KIND_DENSE, VS_KIND_CSR = 0, 1


class VsearchNativeError(RuntimeError):
    pass


class IndexInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("store_dtype", C.c_int32), ("n_rows", C.c_int64), ("n_cols", C.c_int32),
                ("device", C.c_int32), ("nnz", C.c_int64), ("n_packets", C.c_int64), ("device_bytes", C.c_int64),
                ("bytes_per_pass", C.c_int64), ("lanes_per_row", C.c_int32), ("queries_per_pass", C.c_int32),
                ("last_scan_bytes", C.c_int64), ("aux_bytes", C.c_int64), ("last_path", C.c_int32), ("last_fallbacks", C.c_int32),
                ("last_walk_postings", C.c_int64), ("head_columns", C.c_int32), ("postings_state", C.c_int32),
                ("postings_walk", C.c_int32), ("last_packed_tiles", C.c_int32)]


_vp, _i32, _i64, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_int
_SIGNATURES = {
    "vs_version": ([], _int),
    "vs_last_error": ([], C.c_char_p),
    "vs_device_count": ([C.POINTER(_i32)], _int),
    "vs_index_create_csr": ([_vp, _int, _vp, _int, _vp, _int, _int, _i64, _i32, _int, C.POINTER(_vp)], _int),
    "vs_index_create_reserved": ([_i64, _i64, _i32, _int, _int, C.POINTER(_vp)], _int),
    "vs_index_append_csr": ([_vp, _vp, _int, _vp, _int, _vp, _int, _i64], _int),
    "vs_index_slice_rows": ([_vp, _i64, _i64, _int, C.POINTER(_vp)], _int),
    "vs_npz_inspect": ([C.c_char_p, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)], _int),
    "vs_index_append_npz": ([_vp, C.c_char_p, _i32], _int),
    "vs_index_save_npz": ([_vp, C.c_char_p, _int], _int),
    "vs_index_save_native": ([_vp, C.c_char_p], _int),
    "vs_index_load_native": ([C.c_char_p, _int, C.POINTER(_vp)], _int),
    "vs_index_create_dense": ([_vp, _int, _int, _i64, _i32, _i64, _int, C.POINTER(_vp)], _int),
    "vs_index_create_dense_auto": ([_vp, _int, _int, _i64, _i32, _i64, C.c_double, _int, C.POINTER(_vp)], _int),
    "vs_index_create_synthetic": ([C.c_uint64, _i64, _i64, _i32, _i32, _int, _int, _int, _int, C.POINTER(_vp)], _int),
    "vs_index_search": ([_vp, _vp, _int, _i64, _i32, _i32, _i64, _vp, _vp, _vp], _int),
    "vs_index_scores": ([_vp, _vp, _int, _i64, _i32, _vp, _vp], _int),
    "vs_index_prepare": ([_vp, _vp], _int),
    "vs_index_info": ([_vp, C.POINTER(IndexInfo)], _int),
    "vs_index_set_option": ([_vp, C.c_char_p, _int], _int),
    "vs_index_set_queries_per_pass": ([_vp, _int], _int),
    "vs_index_export_csr": ([_vp, _vp, _vp, _vp, _int], _int),
    "vs_index_export_dense": ([_vp, _vp, _int, _i64], _int),
    "vs_index_destroy": ([_vp], None),
    "vs_shard_group_create": ([C.POINTER(_vp), _i32, C.POINTER(_vp)], _int),
    "vs_shard_group_search": ([_vp, _vp, _int, _i64, _i32, _i32, _vp, _vp], _int),
    "vs_shard_group_destroy": ([_vp], None),
    "vs_merge_topk": ([_vp, _vp, _i32, _i64, _i32, _vp, _vp, _int, _vp], _int),
    "vs_topk_mask": ([_vp, _i32, _i32, _i64, _i32, _vp, _int, _vp], _int),
    "vs_bow_mask": ([_vp, _i32, _i32, _i32, _i32, _int, _vp, _int, _vp], _int),
    "vs_embed_mask": ([_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _int, _int, _int, _vp], _int),
    "vs_dense_to_csr": ([_vp, _i32, _i32, _i64, _vp, _vp, _vp, _i64, _int, _vp], _int),
    "vs_embed_mask_to_csr": ([_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _int, _vp, _vp, _vp, _i64, _int, _vp], _int),
    "vs_head_pool": ([_vp, _i32, _i32, _i32, _vp, _int, _vp], _int),
    "vs_head_project_pool": ([_vp, _vp, _i32, _i32, _i32, _i32, _vp, _int, _vp], _int),
    "vs_elu1p": ([_vp, _i64, _vp, _int, _vp], _int),
    "vs_head_pool_mean_topk": ([_vp, _i32, _i32, _i32, _i32, _vp, _int, _vp], _int),
    "vs_rerank_scores": ([_vp, _int, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _int, _vp], _int),
    "vs_rerank_topk": ([_vp, _vp, _i32, _i32, _vp, _vp, _int, _vp], _int),
    "vs_bot_build": ([_vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp], _int),
    "vs_profile_enable": ([_int], _int),
    "vs_profile_reset": ([], _int),
    "vs_profile_read": ([C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i64)], _int),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def build(force: bool = False) -> str:
    """Compile libvsearch_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VsearchNativeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {CSRC}`). vsearch_amd has no CPU fallback.")
        # PyTorch-ROCm ships its own HIP/HSA runtime (torch/lib/libamdhip64.so, same SONAME as
        # /opt/rocm's).  Two runtimes in one process leave the second without devices, so torch's
        # must be loaded first; libvsearch_hip's DT_NEEDED libamdhip64.so.7 then binds to it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the ABI and the header drift apart
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = handle
    return _lib


def last_error() -> str:
    return lib().vs_last_error().decode("utf-8", "replace")


def check(rc: int):
    """Map VS_E* codes to the exception types the reference raises at the same points."""
    if rc == VS_OK:
        return
    msg = last_error()
    if rc == VS_ERANGE:
        raise RuntimeError(msg)                      # torch.topk: "selected index k out of range"
    if rc == VS_EINVAL:
        raise ValueError(msg)
    if rc == VS_ENOMEM:
        raise MemoryError(msg)
    if rc == VS_EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise VsearchNativeError(f"libvsearch_hip error {rc}: {msg}")


def device_count() -> int:
    n = _i32(0)
    check(lib().vs_device_count(C.byref(n)))
    return n.value


def require_device():
    if device_count() <= 0:
        raise VsearchNativeError("no HIP device visible: vsearch_amd computes on MI355X only (no CPU fallback)")
