"""Deterministic synthetic corpora / queries for the vocabulary-space retrieval path.

Counter-based (stateless) generator: every row is a pure function of (seed, row id), so
  * the numpy code below (host, small N: tests, goldens, CPU-baseline sample),
  * ``oracle/vs_oracle.c``  (``vso_synth_*``: fast host twin), and
  * ``vsearch_amd/csrc/synth.hip`` (``vs_index_create_synthetic``: generates straight into the
    device format, 21 M rows x 768 nnz never touch the host)
produce bit-identical rows. Workload law follows SURVEY.md §8(d):
  * VDR-like rows: exactly ``nnz`` distinct columns, uniform without replacement over V, sorted
    (canonical CSR); values in [0.01, 3.01) on a 2^-14 grid (exactly representable fp32).
  * BoT-like rows: binary, length ~ 1 + IrwinHall(4) * 42.3 (mean 85.6, Wiki21M density 0.29 %,
    /root/reference/test/svdr_wiki21m/build_binary_token_index.sh:14-15).
  * queries: ``nnz_q`` (= 768 + 8) distinct columns; values as rows, or dyadic m/64, m in [1,255]
    (exactly summable fp32 -> bit-exact scores on the binary path).

Distinct uniform columns come from a keyed 16-bit Feistel permutation with cycle walking
(a bijection on [0, V)), evaluated at j = 0..nnz-1.
"""
from __future__ import annotations

import numpy as np

U64 = np.uint64
U32 = np.uint32
_M64 = (1 << 64) - 1

KIND_VDR = 0       # fixed nnz per row, fp32 values
KIND_BOT = 1       # binary, variable nnz
KIND_SKEW = 2      # fixed nnz per row, fp32 values, column popularity ~ 1 / rank (Zipf s = 1 with a saturated head)
VAL_GRID = 0       # (164 + h % 49152) / 16384
VAL_DYADIC = 1     # (1 + h % 255) / 64
VAL_ONE = 2        # 1.0


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wraps mod 2^64)."""
    x = np.asarray(x, dtype=U64)
    with np.errstate(over="ignore"):
        z = x + U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)
        return z ^ (z >> U64(31))


def hash2(seed, a):
    with np.errstate(over="ignore"):
        return splitmix64(splitmix64(U64(seed)) ^ (np.asarray(a, dtype=U64) * U64(0xD1342543DE82EF95)))


def hash3(seed, a, b):
    with np.errstate(over="ignore"):
        return splitmix64(hash2(seed, a) + np.asarray(b, dtype=U64) * U64(0x2545F4914F6CDD1D))


def _feistel16(x, key):
    """4-round balanced Feistel on 16-bit x (uint32 arrays); key = uint64 per element."""
    x = x.astype(U32)
    L = x >> U32(8)
    R = x & U32(0xFF)
    for i in range(4):
        k = ((key >> U64(16 * i)) & U64(0xFFFF)).astype(U32)
        with np.errstate(over="ignore"):
            t = ((R ^ k) * U32(0x9E3779B1) + k) & U32(0xFFFFFFFF)
        F = (t >> U32(24)) & U32(0xFF)
        L, R = R, (L ^ F)
    return (L << U32(8)) | R


def perm_cols(key, j, n_cols):
    """Image of j (array, < n_cols) under the keyed bijection on [0, n_cols)."""
    assert 0 < n_cols <= 65536
    x = _feistel16(np.asarray(j, dtype=U32), key)
    bad = x >= n_cols
    while bad.any():
        x = np.where(bad, _feistel16(x, key), x)
        bad = x >= n_cols
    return x


def skew_cols(key, j, lens, n_cols):
    """KIND_SKEW: element j of a row of `lens` non-zeros (arrays) -> column id; twin of synth_skew_col
    (vsearch_amd/csrc/synth_device.h): ranks 1..127 always, equal counts from every octave [2^k, 2^(k+1)) above."""
    j = np.asarray(j, dtype=np.int64)
    lens = np.asarray(lens, dtype=np.int64)
    key = np.asarray(key, dtype=U64)
    head = 127
    kmax = 0
    while (2 << kmax) <= n_cols:
        kmax += 1
    rank = j + 1
    if kmax >= 8:
        gen = (lens > head + 8) & (j >= head)
        if gen.any():
            n_oct = kmax - 7 + 1
            per = (lens[gen] - head) // n_oct
            t = j[gen] - head
            o = np.minimum(t // per, n_oct - 1)
            i = (t - o * per).astype(U64)
            k = 7 + o
            lo = (np.int64(1) << k)
            size = np.where((np.int64(2) << k) <= n_cols + 1, lo, n_cols + 1 - lo)
            mask = (lo - 1).astype(U64)
            with np.errstate(over="ignore"):
                h = splitmix64(key[gen] + U64(0x9E3779B97F4A7C15) * (k + 1).astype(U64))
            odd = (h & U64(0xFFFFFFFF)) | U64(1)
            add = h >> U64(32)
            x = i.copy()
            todo = np.ones(x.shape, dtype=bool)
            while todo.any():
                with np.errstate(over="ignore"):
                    nx = ((x * odd + add) & U64(0xFFFFFFFF)) & mask        # 32-bit arithmetic, then the octave mask
                x = np.where(todo, nx, x)
                todo = todo & (x >= size.astype(U64))
            rank = rank.copy()
            rank[gen] = lo + x.astype(np.int64)
    gkey = np.full(rank.shape, 0x5A495046534B4557, dtype=U64)
    return perm_cols(gkey, (rank - 1).astype(U32), n_cols)


def row_lengths(seed, rows, kind, nnz):
    rows = np.asarray(rows, dtype=np.int64)
    if kind in (KIND_VDR, KIND_SKEW):
        return np.full(rows.shape, nnz, dtype=np.int64)
    h = hash3(seed, rows, 0x4C454E)  # "LEN"
    # 4 x 16-bit uniforms -> Irwin-Hall; mean = 1 + 4*0.5*(nnz-1)*0.5 ... scaled so mean == nnz
    s = np.zeros(rows.shape, dtype=np.int64)
    for i in range(4):
        s += ((h >> U64(16 * i)) & U64(0xFFFF)).astype(np.int64)
    # s in [0, 4*65535]; len = 1 + floor(s * (nnz-1) / (2*65536))  (mean ~= nnz)
    return 1 + (s * (int(nnz) - 1)) // (2 * 65536)


def values_for(seed, rows, cols, val_law):
    h = hash3(seed ^ 0x56414C, rows, cols)  # "VAL"
    if val_law == VAL_GRID:
        m = (h % U64(49152)).astype(np.float32)
        return ((np.float32(164.0) + m) / np.float32(16384.0)).astype(np.float32)
    if val_law == VAL_DYADIC:
        m = (h % U64(255)).astype(np.float32)
        return ((np.float32(1.0) + m) / np.float32(64.0)).astype(np.float32)
    return np.ones(np.shape(h), dtype=np.float32)


def synth_csr(seed, row0, n_rows, n_cols=29523, nnz=768, kind=KIND_VDR, val_law=VAL_GRID):
    """Rows [row0, row0+n_rows) of the synthetic matrix as canonical CSR.

    Returns (indptr int64 [n_rows+1], indices int32 [nnz_total], data float32 [nnz_total]).
    """
    rows = np.arange(row0, row0 + n_rows, dtype=np.int64)
    lens = row_lengths(seed, rows, kind, nnz)
    lens = np.minimum(lens, n_cols)
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(lens, out=indptr[1:])
    total = int(indptr[-1])
    row_of = np.repeat(rows, lens)
    j = np.arange(total, dtype=np.int64) - np.repeat(indptr[:-1], lens)
    key = hash3(seed, row_of, 0x4B4559)  # "KEY"
    if kind == KIND_SKEW:
        cols = skew_cols(key, j, np.repeat(lens, lens), n_cols).astype(np.int64)
    else:
        cols = perm_cols(key, j, n_cols).astype(np.int64)
    # sort columns within each row (canonical CSR): composite key sort
    order = np.argsort(row_of * 65536 + cols, kind="stable")
    cols = cols[order]
    if kind == KIND_BOT or val_law == VAL_ONE:
        data = np.ones(total, dtype=np.float32)
    else:
        data = values_for(seed, row_of, cols, val_law)
    return indptr, cols.astype(np.int32), data


def synth_queries(seed, n_q, n_cols=29523, nnz_q=776, val_law=VAL_GRID, q0=0, kind=KIND_VDR):
    """Dense [n_q, n_cols] float32 query matrix, nnz_q non-zeros per row (kind: KIND_VDR | KIND_SKEW column law)."""
    indptr, cols, data = synth_csr(seed, q0, n_q, n_cols, nnz_q, kind, val_law)
    q = np.zeros((n_q, n_cols), dtype=np.float32)
    rows = np.repeat(np.arange(n_q), np.diff(indptr))
    q[rows, cols] = data
    return q


def dense_uniform(seed, shape, lo=-1.0, hi=1.0):
    """fp32 uniform [lo, hi) on a 2^-24 grid; element i of the flattened array = f(seed, i)."""
    n = int(np.prod(shape))
    h = hash2(seed, np.arange(n, dtype=np.int64))
    u = ((h >> U64(40)).astype(np.float64)) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def dense_tiefree(seed, shape, lo=-3.0, hi=5.0):
    """fp32 rows with all-distinct entries (row-wise permutation of an arithmetic grid): top-k has no ties."""
    rows, v = shape
    out = np.empty(shape, dtype=np.float32)
    for r in range(rows):
        h = hash3(seed, r, np.arange(v, dtype=np.int64))
        rank = np.empty(v, dtype=np.int64)
        rank[np.argsort(h, kind="stable")] = np.arange(v)
        out[r] = (lo + (hi - lo) * (rank.astype(np.float64) + 0.5) / v).astype(np.float32)
    return out
