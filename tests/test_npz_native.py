"""The library's own reader of scipy.sparse.save_npz shards (csrc/npz.hip) against scipy: stored and deflated archives, int32 and
int64 indices, fp32 / fp64 / fp16-widened values, unsorted rows, the column shift (SparseIndex.init_index: `load_npz(f)[:, shift:]`,
reference index.py:172).  vs_npz_inspect needs no GPU; the append path is checked on the GPU."""
import numpy as np
import pytest
import scipy.sparse as sp

from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, npz_inspect


def _random_csr(rng, n, m, density, dtype=np.float32, index_dtype=np.int32, sort=True):
    mat = sp.random(n, m, density=density, format="csr", random_state=rng, dtype=np.float64)
    mat.data = (0.25 + rng.random(mat.nnz)).astype(dtype)
    mat = sp.csr_matrix((mat.data, mat.indices.astype(index_dtype), mat.indptr.astype(index_dtype)), shape=(n, m))
    if not sort:                                                       # reverse every row's column order
        for r in range(n):
            a, b = mat.indptr[r], mat.indptr[r + 1]
            mat.indices[a:b] = mat.indices[a:b][::-1].copy()
            mat.data[a:b] = mat.data[a:b][::-1].copy()
        mat.has_sorted_indices = False
    return mat


@pytest.mark.parametrize("compressed", [True, False], ids=["deflate", "stored"])
@pytest.mark.parametrize("index_dtype", [np.int32, np.int64], ids=["i4", "i8"])
def test_inspect_matches_scipy(tmp_path, compressed, index_dtype):
    rng = np.random.default_rng(5)
    mat = _random_csr(rng, 300, 1500, 0.03, index_dtype=index_dtype)
    path = str(tmp_path / "shard.npz")
    sp.save_npz(path, mat, compressed=compressed)
    for shift in (0, 1, 999):
        cut = mat[:, shift:]
        lens = np.diff(cut.indptr)
        assert npz_inspect(path, shift) == (300, 1500 - shift, int(cut.nnz), int(((lens + 7) // 8).sum()))


def test_inspect_rejects_other_files(tmp_path):
    rng = np.random.default_rng(6)
    mat = _random_csr(rng, 50, 80, 0.1)
    p_coo = str(tmp_path / "coo.npz")
    sp.save_npz(p_coo, mat.tocoo())
    with pytest.raises(NotImplementedError):
        npz_inspect(p_coo, 0)
    p_bad = str(tmp_path / "bad.npz")
    with open(p_bad, "wb") as f:
        f.write(b"this is not a zip archive at all" * 8)
    with pytest.raises(ValueError):
        npz_inspect(p_bad, 0)
    p_csr = str(tmp_path / "ok.npz")
    sp.save_npz(p_csr, mat)
    raw = open(p_csr, "rb").read()
    p_cut = str(tmp_path / "cut.npz")
    with open(p_cut, "wb") as f:
        f.write(raw[: len(raw) // 2])
    with pytest.raises(ValueError):
        npz_inspect(p_cut, 0)
    with pytest.raises(ValueError):
        npz_inspect(p_csr, 81)                                         # shift beyond the columns


def _write_npz(path, **arrays):
    """An .npz with arbitrary (possibly inconsistent) members, written member by member so that headers can be doctored."""
    import io, zipfile
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        for name, a in arrays.items():
            if isinstance(a, bytes):
                z.writestr(name + ".npy", a)
            else:
                buf = io.BytesIO()
                np.save(buf, a)
                z.writestr(name + ".npy", buf.getvalue())


def test_inspect_survives_hostile_files(tmp_path):
    """the reader must treat the file as untrusted -- a row pointer far beyond the indices, an item size it cannot read,
    2-d members, size fields that would allocate the world: every one an error code (ValueError / NotImplementedError), no read past
    a buffer, no C++ exception through the C ABI (which would kill the process)."""
    import io
    fmt = np.array("csr")
    shape = np.array([2, 10], dtype=np.int64)
    good_idx = np.array([1, 2, 3, 4, 5], dtype=np.int32)
    good_dat = np.ones(5, dtype=np.float32)
    # (1) indptr = [0, 1e9, 5]: consistent at both ends, wild in the middle (reproduced by the advisor as a heap read)
    p = str(tmp_path / "spike.npz")
    _write_npz(p, indices=good_idx, indptr=np.array([0, 10**9, 5], dtype=np.int64), format=fmt, shape=shape, data=good_dat)
    with pytest.raises(ValueError):
        npz_inspect(p, 0)
    # (2) a negative row pointer
    p = str(tmp_path / "neg.npz")
    _write_npz(p, indices=good_idx, indptr=np.array([0, -3, 5], dtype=np.int64), format=fmt, shape=shape, data=good_dat)
    with pytest.raises(ValueError):
        npz_inspect(p, 0)
    # (3) item sizes the integer reader does not know: '<i3', '<i0' (header doctored by hand)
    buf = io.BytesIO()
    np.save(buf, np.array([0, 2, 5], dtype=np.int32))
    raw = buf.getvalue()
    for bad in (b"<i3", b"<i0", b"<i9"):
        p = str(tmp_path / ("descr_" + bad.decode()[1:] + ".npz"))
        _write_npz(p, indices=good_idx, indptr=raw.replace(b"<i4", bad), format=fmt, shape=shape, data=good_dat)
        with pytest.raises((ValueError, NotImplementedError)):
            npz_inspect(p, 0)
    # (4) 2-d indptr / indices
    p = str(tmp_path / "twod.npz")
    _write_npz(p, indices=good_idx.reshape(5, 1), indptr=np.array([[0, 2, 5]], dtype=np.int64).reshape(3, 1), format=fmt, shape=shape, data=good_dat)
    with pytest.raises(ValueError):
        npz_inspect(p, 0)
    # (5) a shape whose element count overflows int64
    buf = io.BytesIO()
    np.save(buf, np.zeros(3, dtype=np.int64))
    raw = buf.getvalue().replace(b"(3,)", b"(4611686018427387904, 4611686018427387904)"[:0] + b"(3,)")   # (keep the header length: patch below)
    hdr_pad = raw.index(b"(3,)")
    big = b"(9223372036854775807,9)"
    doctored = raw[:hdr_pad] + big + raw[hdr_pad + 4 + (len(big) - 4):]                                    # overwrite padding spaces
    p = str(tmp_path / "overflow.npz")
    _write_npz(p, indices=good_idx, indptr=doctored, format=fmt, shape=shape, data=good_dat)
    with pytest.raises((ValueError, NotImplementedError)):
        npz_inspect(p, 0)
    # (6) a central directory that claims to be larger than the file
    ok = str(tmp_path / "ok.npz")
    sp.save_npz(ok, sp.csr_matrix((good_dat, good_idx, np.array([0, 2, 5])), shape=(2, 10)), compressed=False)
    raw = bytearray(open(ok, "rb").read())
    eocd = raw.rfind(b"PK\x05\x06")
    raw[eocd + 12:eocd + 16] = (0x7FFFFFF0).to_bytes(4, "little")
    p = str(tmp_path / "cdsize.npz")
    open(p, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        npz_inspect(p, 0)
    # the untouched file still reads
    assert npz_inspect(ok, 0) == (2, 10, 5, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("store", [nat.VS_F32, nat.VS_F16, nat.VS_NONE], ids=["fp32", "fp16", "binary"])
def test_append_npz_equals_scipy_slice(tmp_path, store):
    rng = np.random.default_rng(7)
    shift = 37
    shards = [_random_csr(rng, 700, 4000, 0.02, dtype=np.float64, index_dtype=np.int64, sort=False),
              _random_csr(rng, 450, 4000, 0.05, dtype=np.float32)]
    if store == nat.VS_NONE:
        for m in shards:
            m.data[:] = 1
    paths = []
    for i, m in enumerate(shards):
        paths.append(str(tmp_path / f"s{i}.npz"))
        sp.save_npz(paths[-1], m, compressed=bool(i))
    sizes = [npz_inspect(p, shift) for p in paths]
    idx = DeviceIndex.reserved(sum(s[0] for s in sizes), sum(s[3] for s in sizes), 4000 - shift, store)
    for p in paths:
        idx.append_npz(p, shift)
    ip, ix, d = idx.export_csr(np.float32)
    ref = sp.vstack([m.tocsr()[:, shift:] for m in shards]).tocsr()
    ref.sort_indices()
    assert (ip == ref.indptr).all() and (ix == ref.indices).all()
    want = ref.data.astype(np.float32)
    if store == nat.VS_F16:
        want = want.astype(np.float16).astype(np.float32)
    assert (d == want).all()
    if store == nat.VS_NONE:
        shards[0].data[3] = 2.0
        sp.save_npz(paths[0], shards[0])
        idx2 = DeviceIndex.reserved(sizes[0][0], sizes[0][3] + 8, 4000 - shift, store)
        with pytest.raises(ValueError):
            idx2.append_npz(paths[0], shift)


@pytest.mark.gpu
@pytest.mark.parametrize("compressed", [False, True], ids=["stored", "deflate"])
@pytest.mark.parametrize("store", [nat.VS_F32, nat.VS_NONE], ids=["fp32", "binary"])
def test_save_npz_is_read_by_scipy_and_by_the_library(tmp_path, compressed, store):
    """vs_index_save_npz writes what scipy.sparse.load_npz reads (SparseIndex.save, reference index.py:181-202) and what the
    library's own reader reads back."""
    rng = np.random.default_rng(11)
    mat = _random_csr(rng, 900, 3000, 0.03, dtype=np.float32, index_dtype=np.int64)
    if store == nat.VS_NONE:
        mat.data[:] = 1
    idx = DeviceIndex.from_csr(mat.indptr, mat.indices, None if store == nat.VS_NONE else mat.data, 3000, store_dtype=store)
    path = str(tmp_path / "out.npz")
    idx.save_npz(path, compressed=compressed)
    back = sp.load_npz(path)
    assert back.format == "csr" and back.shape == (900, 3000)
    assert back.indices.dtype == np.int64 and back.indptr.dtype == np.int64 and back.data.dtype == np.float32
    assert (back.indptr == mat.indptr).all() and (back.indices == mat.indices).all() and (back.data == mat.data).all()
    with np.load(path) as z:
        assert sorted(z.files) == ["_is_array", "data", "format", "indices", "indptr", "shape"] and z["format"].item() == b"csr"
    n_r, n_c, nnz, pk = npz_inspect(path, 0)
    again = DeviceIndex.reserved(n_r, pk, n_c, store)
    again.append_npz(path, 0)
    ip, ix, d = again.export_csr(np.float32)
    assert (ip == mat.indptr).all() and (ix == mat.indices).all() and (d == mat.data).all()
