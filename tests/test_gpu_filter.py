"""GPU parity tests of the filter-and-refine postings search (bp_walk.h / bp_refine.h) -- run on MI355X.

The search is exact by construction: the fixed-point walk only proposes candidates, the refine step re-scores them with the
library's exact numerics and proves the top k, unproven queries take an exact pass.  Every mode below must therefore return
the CSR scan's ids and scores BIT FOR BIT (same canonical order), and a valid top-k of the oracle's score matrix."""
import numpy as np
import pytest

import oracle
from oracle import compare
from conftest import V
from vsearch_amd import synth
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _search(idx, q, k, **opts):
    for name, value in opts.items():
        idx.set_option(name, value)
    ids, sc = idx.search(q, k)
    return np.asarray(ids), np.asarray(sc), idx.info()



def _oracle_pin(seed, n, q, ids, sc, k, kind=0, val_law=0, nnz=768, windows=48, window_rows=8192, binary=False, exact=False):
    """Pin a result on an index too large for the oracle to the CPU oracle: the synthetic rows are a pure function of (seed, row), so
    the oracle (a) regenerates every RETURNED row and re-scores it (fp64 sums): the returned scores must be the oracle's, and (b)
    regenerates `windows` runs of `window_rows` consecutive rows spread over the whole index and scores them all: no sampled row
    may beat a query's k-th returned score unless it was returned."""
    ids = np.asarray(ids); sc = np.asarray(sc)
    B = q.shape[0]
    for b in range(B):
        rows = ids[b]
        parts = [oracle.synth_csr(seed, int(r), 1, V, nnz, kind, val_law) for r in rows]
        ip = np.concatenate([[0], np.cumsum([len(p[1]) for p in parts])]).astype(np.int64)
        ix = np.concatenate([p[1] for p in parts]); d = None if binary else np.concatenate([p[2] for p in parts])
        _, _, allsc = oracle.csr_search(ip, ix, d, V, q[b:b + 1], 1, acc64=True, return_all=True)
        want = allsc[0].astype(np.float32)
        if exact:
            assert (want == sc[b]).all(), f"query {b}: returned scores differ from the oracle's"
        else:
            assert np.allclose(want, sc[b], rtol=RTOL, atol=0), f"query {b}: returned scores differ from the oracle's by {np.abs(want - sc[b]).max()}"
    starts = np.linspace(0, n - window_rows, windows).astype(np.int64)
    # ... plus windows that STRADDLE block boundaries of the column-grouped copy (1920- and 2048-document blocks: a window of 8192+ rows
    # crosses four of them wherever it starts -- these are centred on a boundary in the middle and near the end) and the index's LAST rows
    # (the last, partial block)
    extra = [max(0, (n // 2 // 1920) * 1920 - window_rows // 2), max(0, (n // 2 // 2048) * 2048 - window_rows // 2),
             max(0, ((n - 1) // 1920) * 1920 - window_rows // 2), max(0, n - window_rows)]
    starts = np.unique(np.concatenate([starts, np.minimum(np.asarray(extra, dtype=np.int64), n - window_rows)]))
    for r0 in starts:
        ip, ix, d = oracle.synth_csr(seed, int(r0), window_rows, V, nnz, kind, val_law)
        _, _, allsc = oracle.csr_search(ip, ix, None if binary else d, V, q, 1, acc64=True, return_all=True)
        for b in range(B):
            kth = sc[b, k - 1]
            better = np.nonzero(allsc[b] > kth * (1 + RTOL if kth > 0 else 1 - RTOL))[0] + r0
            missing = np.setdiff1d(better, ids[b])
            assert missing.size == 0, f"query {b}: rows {missing[:5]} beat the k-th score {kth} and were not returned"


def _experimental_walks(idx):
    """the experimental walks of round 3 (postings_walk = 1 .. 3) are compiled only with `make EXPERIMENTAL=1`"""
    try:
        idx.set_option("postings_walk", 1)
    except Exception:
        return False
    idx.set_option("postings_walk", -1)
    return True


def _modes(idx, q, k, want_quant, tied_queries=0):
    """-> results of {csr scan, filter on the default walk (quad chunks where they apply), filter on the list walk, both on exact records,
    forced fallback x 2, the experimental walks when the library has them, fp64 walk}; checks paths and bit-equality."""
    ref_ids, ref_sc, info = _search(idx, q, k, blocked_postings=0)
    assert info.last_path == 1
    out = {}
    modes = [("filter", dict(postings_filter=1, postings_quant=-1, postings_force_fallback=0, postings_walk=-1)),
             ("filter-quad-walk", dict(postings_filter=1, postings_quant=-1, postings_force_fallback=0, postings_walk=4)),
             ("filter-list-walk", dict(postings_filter=1, postings_quant=-1, postings_force_fallback=0, postings_walk=0)),
             ("filter-exact-records", dict(postings_filter=1, postings_quant=0, postings_force_fallback=0, postings_walk=-1)),
             ("fallback", dict(postings_filter=1, postings_quant=-1, postings_force_fallback=1, postings_walk=-1)),
             ("fallback-list-walk", dict(postings_filter=1, postings_quant=-1, postings_force_fallback=1, postings_walk=0)),
             ("fallback-exact-records", dict(postings_filter=1, postings_quant=0, postings_force_fallback=1, postings_walk=-1))]
    if _experimental_walks(idx):
        modes += [("filter-flat-walk", dict(postings_filter=1, postings_quant=-1, postings_force_fallback=0, postings_walk=1)),
                  ("filter-flat-walk-exact-records", dict(postings_filter=1, postings_quant=0, postings_force_fallback=0, postings_walk=1)),
                  ("filter-pipe-walk", dict(postings_filter=1, postings_quant=-1, postings_force_fallback=0, postings_walk=2)),
                  ("filter-pipe-walk-exact-records", dict(postings_filter=1, postings_quant=0, postings_force_fallback=0, postings_walk=2)),
                  ("filter-stream-walk", dict(postings_filter=1, postings_quant=-1, postings_force_fallback=0, postings_walk=3)),
                  ("filter-stream-walk-exact-records", dict(postings_filter=1, postings_quant=0, postings_force_fallback=0, postings_walk=3))]
    modes += [("fp64-walk", dict(postings_filter=0, postings_force_fallback=0, postings_walk=-1))]
    for name, opts in modes:
        ids, sc, info = _search(idx, q, k, blocked_postings=1, **opts)
        if name == "fp64-walk":
            assert info.last_path == 2
        else:
            assert info.last_path == 3
            assert info.last_fallbacks == (q.shape[0] if name.startswith("fallback") else tied_queries), name
        assert (ids == ref_ids).all() and (sc == ref_sc).all(), f"{name}: differs from the CSR scan"
        out[name] = info
    idx.set_option("postings_force_fallback", 0)
    idx.set_option("postings_quant", -1)
    idx.set_option("postings_filter", 1)
    idx.set_option("postings_walk", -1)
    return ref_ids, ref_sc


@pytest.mark.parametrize("store", [nat.VS_F32, nat.VS_F16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("k", [1, 100, 700])
def test_filter_refine_is_bit_identical_to_the_csr_scan(store, k):
    n = 6000
    ip, ix, d = oracle.synth_csr(0, 0, n)
    if store == nat.VS_F16:
        d = d.astype(np.float16).astype(np.float32)
    idx = DeviceIndex.from_csr(ip, ix, d, V, store_dtype=store)
    law = synth.VAL_GRID if store == nat.VS_F32 else synth.VAL_DYADIC
    q = oracle.synth_queries(1, 19, val_law=law)                  # ragged batch: 2 full tiles + 3 queries
    ids, sc = _modes(idx, q, k, want_quant=store == nat.VS_F32)
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q, k, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL)


def test_filter_on_ragged_rows_and_partial_last_block():
    """Rows of 0..2000 non-zeros, 2 blocks + a partial one, empty rows, an all-zero query and a one-column query."""
    rng = np.random.default_rng(7)
    n = 2 * 2048 + 777
    lens = rng.integers(0, 300, size=n)
    lens[::131] = 2000
    lens[[0, 5, n - 1]] = 0
    ip = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=ip[1:])
    ix = np.concatenate([np.sort(rng.choice(V, size=l, replace=False)) for l in lens]).astype(np.int32)
    d = (0.01 + 3 * rng.random(len(ix))).astype(np.float32)
    q = oracle.synth_queries(3, 10)
    q[4] = 0.0
    q[7] = 0.0
    q[7, ix[10]] = 2.5
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    ids, sc = _modes(idx, q, 50, want_quant=True, tied_queries=1)      # the one-column query: thousands of documents tie at score 0
    _, _, allsc = oracle.csr_search(ip, ix, d, V, q, 50, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)


def test_ties_beyond_the_candidate_margin_take_the_exact_pass():
    """600 identical best documents: no candidate set of k + 28 can be proven, so those queries must fall back -- and still
    return the canonical (lowest ids first) answer."""
    ip, ix, d = oracle.synth_csr(2, 0, 5000)
    ix2, d2 = ix.copy(), d.copy()
    q = oracle.synth_queries(5, 8)
    cols = np.nonzero(q[0])[0][:768].astype(ix.dtype)
    for r in range(100, 700):                                      # rows 100..699: the same row, hitting query 0 on every column
        ix2[ip[r]:ip[r + 1]] = np.sort(cols)
        d2[ip[r]:ip[r + 1]] = 2.0
    idx = DeviceIndex.from_csr(ip, ix2, d2, V)
    ref_ids, ref_sc, _ = _search(idx, q, 100, blocked_postings=0)
    ids, sc, info = _search(idx, q, 100, blocked_postings=1)
    assert info.last_path == 3 and info.last_fallbacks >= 1
    assert (ids == ref_ids).all() and (sc == ref_sc).all()
    assert (ids[0] == np.arange(100, 200)).all()
    ids, sc, info = _search(idx, q, 100, blocked_postings=1, postings_quant=0)
    assert info.last_fallbacks >= 1 and (ids == ref_ids).all() and (sc == ref_sc).all()


@pytest.mark.parametrize("quant", [-1, 0], ids=["lossy-records", "exact-records"])
def test_short_and_dominant_weight_queries(quant):
    """Queries of 1, 2, 8 and 31 non-zeros and queries with one dominant weight: the per-query scale then puts single products
    above 2^25 fixed-point units, where the fp32 rounding of a product exceeds the one unit of truncation -- the proof's slack has to
    cover it (bp_refine.h).  Index values sit at fp16 / fp32 rounding midpoints, and blocks of duplicated rows make near-ties at the
    k-th rank.  Results: the CSR scan's, bit for bit, and a valid top-k of the oracle's fp64 scores."""
    rng = np.random.default_rng(11)
    n = 20000
    ip, ix, d = oracle.synth_csr(5, 0, n)
    d = d.copy()
    # values at rounding midpoints: halfway between two fp16 numbers (exactly representable in fp32), and fp32 numbers with a long tail
    h = d[::3].astype(np.float16).astype(np.float32)
    d[::3] = h * (1 + 2.0 ** -11)
    d[1::3] = np.nextafter(d[1::3].astype(np.float16).astype(np.float32), np.float32(4))
    # near-ties: 300 rows equal row 7 except for one value nudged by an ulp
    for j, r in enumerate(range(1000, 1300)):
        ix[ip[r]:ip[r + 1]] = ix[ip[7]:ip[8]]
        d[ip[r]:ip[r + 1]] = d[ip[7]:ip[8]]
        d[ip[r] + (j % 768)] = np.nextafter(d[ip[r] + (j % 768)], np.float32(4))
    cols7 = ix[ip[7]:ip[8]]
    qs = []
    for nnz_q in (1, 2, 8, 31):
        for rep in range(2):
            v = np.zeros(V, np.float32)
            c = rng.choice(cols7, size=nnz_q, replace=False) if rep == 0 else rng.choice(V, size=nnz_q, replace=False)
            v[c] = (0.01 + 3 * rng.random(nnz_q)).astype(np.float32)
            qs.append(v)
    for dom in (1e3, 1e6):                                          # 776 non-zeros, one of them dominant
        v = oracle.synth_queries(9, 1)[0].copy()
        v[cols7[5]] = np.float32(dom)
        qs.append(v)
    q = np.stack(qs)
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    ref_ids, ref_sc, info = _search(idx, q, 100, blocked_postings=0)
    assert info.last_path == 1
    for walk in (-1, 4, 0) + ((1, 2, 3) if _experimental_walks(idx) else ()):
        ids, sc, info = _search(idx, q, 100, blocked_postings=1, postings_quant=quant, postings_walk=walk)
        assert info.last_path == 3
        assert (ids == ref_ids).all() and (sc == ref_sc).all(), f"walk {walk}: differs from the CSR scan"
    idx.set_option("postings_walk", -1)
    _, _, allsc = oracle.csr_search(ip, ix, d, V, q, 100, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ref_ids, ref_sc, rtol=RTOL)


def test_signed_values_and_weights():
    """Negative index values keep fp32 records (the lossy copy needs non-negative data); a negative query weight on a lossy
    copy sends that query to the exact pass.  Results stay those of the CSR scan."""
    ip, ix, d = oracle.synth_csr(4, 0, 5000)
    rng = np.random.default_rng(1)
    q = oracle.synth_queries(6, 9)
    qneg = q.copy()
    qneg[2, np.nonzero(q[2])[0][::3]] *= -1.0
    idx = DeviceIndex.from_csr(ip, ix, d, V)                       # non-negative values: lossy copy
    ref = _search(idx, qneg, 100, blocked_postings=0)
    got = _search(idx, qneg, 100, blocked_postings=1)
    assert got[2].last_path == 3 and got[2].last_fallbacks == 1
    assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all()
    dneg = d * np.where(rng.random(len(d)) < 0.3, -1.0, 1.0).astype(np.float32)
    idx2 = DeviceIndex.from_csr(ip, ix, dneg, V)                   # signed values: exact records, signed fixed-point sums
    ref = _search(idx2, qneg, 100, blocked_postings=0)
    got = _search(idx2, qneg, 100, blocked_postings=1)
    assert got[2].last_path == 3
    assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all()
    _, _, allsc = oracle.csr_search(ip, ix, dneg, V, qneg, 100, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, got[0], got[1], rtol=RTOL)


@pytest.mark.parametrize("nnz,rows,chunks", [(86, 0, 6), (200, 0, 6), (600, 0, 6), (600, 2048, 6), (600, 8192, 5)],
                         ids=["short-lists", "four-slot-tiles", "small-blocks", "linked-lists", "long-lists"])
@pytest.mark.parametrize("law", [synth.VAL_DYADIC, synth.VAL_GRID], ids=["dyadic", "fp32-weights"])
def test_binary_index_on_the_postings_walk(law, nnz, rows, chunks):
    """Bag-of-token index (no values): dyadic weights are exact in fixed point (nothing to prove, scores and ids bit-equal to the
    oracle); arbitrary fp32 weights go through the refine step.  The default copy is the bag-of-token chunks (bp_bq.h, postings_walk
    6): 86 tokens a document -> blocks of 6144 documents, two query slots; 200 -> 2048 documents, four slots; 600 -> 768 documents;
    600 tokens in blocks of 2048: lists of ~42 postings, EVERY list goes on in a linked overflow chunk; in blocks of 8192: 150 k
    overflow chunks a block, more than a link's 15 bits address -- the build keeps the records of bp_bin.h (5), whose walk then takes
    its path for lists beyond the two prefetched records on every list.  The records (5) and the list walk (0) are checked as well."""
    n = 30_000 if nnz == 86 else 12_000
    ip, ix, _ = oracle.synth_csr(3, 0, n, V, nnz, synth.KIND_BOT)
    q = oracle.synth_queries(8, 21, val_law=law)
    idx = DeviceIndex.from_csr(ip, ix, None, V)
    ref = _search(idx, q, 100, blocked_postings=0)
    got = _search(idx, q, 100, blocked_postings=1, postings_rows=rows)
    assert got[2].last_path == 3 and got[2].aux_bytes > 0 and got[2].postings_walk == chunks
    assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all()
    for walk, kind in ((5, 5), (0, 0), (6, chunks)):
        lw = _search(idx, q, 100, blocked_postings=1, postings_walk=walk)
        assert lw[2].last_path == 3 and lw[2].postings_walk == kind and (lw[0] == ref[0]).all() and (lw[1] == ref[1]).all(), walk
    idx.set_option("postings_walk", -1)
    forced = _search(idx, q, 100, blocked_postings=1, postings_force_fallback=1)
    assert forced[2].last_fallbacks == q.shape[0]
    assert (forced[0] == ref[0]).all() and (forced[1] == ref[1]).all()
    o_ids, o_sc = oracle.csr_search(ip, ix, None, V, q, 100)
    if law == synth.VAL_DYADIC:
        assert (got[0] == o_ids).all() and (got[1] == o_sc).all()
    else:
        compare.compare_topk(o_ids, o_sc, got[0], got[1], rtol=RTOL)


@pytest.mark.parametrize("n,nnz,rows", [(3_000, 20, 0), (70_000, 40, 0), (20_000, 86, 256), (20_000, 86, 1024), (40_000, 86, 4096), (40_000, 86, 4160),
                                        (40_000, 86, 8192), (9_000, 300, 0)])
def test_bag_of_token_chunks_block_shapes(n, nnz, rows):
    """The bag-of-token chunks (bp_bq.h) over their block shapes: two query slots (blocks above 4096 documents) and four, blocks from
    256 to 8192 documents (8192 at 86 tokens: 24 postings a list, 1 list in 18 goes on in a linked overflow chunk), a partial last
    block, batches of 1 .. 33 queries (tiles with fewer queries than slots), k from 1 to 500 -- and the candidate buffer's overflow
    path: with k = 500 a work item's first block pushes thousands of candidates per slot into 4096 places.  Bit-equal to the CSR scan,
    dyadic weights bit-equal to the oracle."""
    ip, ix, _ = oracle.synth_csr(11, 0, n, V, nnz, synth.KIND_BOT)
    idx = DeviceIndex.from_csr(ip, ix, None, V)
    idx.set_option("postings_rows", rows)
    for B, k in ((1, 1), (3, 500), (7, 100), (33, 10)):
        q = oracle.synth_queries(12 + B, B, val_law=synth.VAL_DYADIC)
        ref = _search(idx, q, k, blocked_postings=0)
        got = _search(idx, q, k, blocked_postings=1)
        assert got[2].last_path == 3 and got[2].postings_walk == 6, (B, k)
        assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all(), (B, k)
        if B == 3:
            o_ids, o_sc = oracle.csr_search(ip, ix, None, V, q, k)
            assert (got[0] == o_ids).all() and (got[1] == o_sc).all()


def test_bag_of_token_packed_sums():
    """Four query slots a tile on packed 16-bit sums (bp_bq.h, option "postings_packed"): queries whose weights are integers at a small
    power-of-two scale with longest row x largest weight < 65 536 take the packed walk; the others -- weights that are too large, weights
    that are no dyadic numbers, a negative weight -- the two-slot int32 walk; a tile with one such query is cut in two.  Whatever the
    mix, ids and scores equal the CSR scan's bit for bit, dyadic batches equal the oracle's."""
    n = 40_000
    ip, ix, _ = oracle.synth_csr(5, 0, n, V, 86, synth.KIND_BOT)
    idx = DeviceIndex.from_csr(ip, ix, None, V)
    for B, k in ((1, 10), (3, 100), (4, 100), (5, 500), (33, 100)):
        q = oracle.synth_queries(40 + B, B, val_law=synth.VAL_DYADIC)
        ref = _search(idx, q, k, blocked_postings=0)
        got = _search(idx, q, k, blocked_postings=1, postings_packed=1)
        assert got[2].last_path == 3 and got[2].postings_walk == 6 and got[2].last_packed_tiles == (B + 3) // 4, (B, k, got[2].last_packed_tiles)
        assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all(), (B, k)
        off = _search(idx, q, k, blocked_postings=1, postings_packed=0)
        assert off[2].last_packed_tiles == 0 and (off[0] == ref[0]).all() and (off[1] == ref[1]).all(), (B, k)
        o_ids, o_sc = oracle.csr_search(ip, ix, None, V, q, k)
        assert (got[0] == o_ids).all() and (got[1] == o_sc).all()
    idx.set_option("postings_packed", -1)
    # a common factor changes nothing (the scale is the smallest power of two that makes the weights integers: here 2^-4) ...
    q = oracle.synth_queries(77, 9, val_law=synth.VAL_DYADIC)
    big = (q * 1024.0).astype(np.float32)
    ref = _search(idx, big, 100, blocked_postings=0)
    got = _search(idx, big, 100, blocked_postings=1)
    assert got[2].last_packed_tiles == 3 and (got[0] == ref[0]).all() and (got[1] == ref[1]).all()
    # ... one weight 8192 x the others does (longest row x largest integer weight >> 65 536): every tile on the int32 walk
    for b in range(9):
        big[b, np.nonzero(big[b])[0][b]] *= 8192.0
    ref = _search(idx, big, 100, blocked_postings=0)
    got = _search(idx, big, 100, blocked_postings=1)
    assert got[2].last_packed_tiles == 0 and (got[0] == ref[0]).all() and (got[1] == ref[1]).all()
    # ... weights with 14 fractional bits (integers up to 49 315): int32 walk
    grid = oracle.synth_queries(78, 9, val_law=synth.VAL_GRID)
    ref = _search(idx, grid, 100, blocked_postings=0)
    got = _search(idx, grid, 100, blocked_postings=1)
    assert got[2].last_packed_tiles == 0 and (got[0] == ref[0]).all() and (got[1] == ref[1]).all()
    # a mix: queries 2 and 7 of 11 do not qualify (too large / a negative weight) -> tiles {0..3} and {4..7} are cut in two, {8..10} is packed
    mix = oracle.synth_queries(79, 11, val_law=synth.VAL_DYADIC)
    mix[2, np.nonzero(mix[2])[0][5]] *= 8192.0
    nzc = np.nonzero(mix[7])[0]
    mix[7, nzc[0]] = -mix[7, nzc[0]]
    ref = _search(idx, mix, 100, blocked_postings=0)
    got = _search(idx, mix, 100, blocked_postings=1)
    assert got[2].last_packed_tiles == 1, got[2].last_packed_tiles
    assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all()
    # the bound is on what a document CAN reach: weights of 255 / 64 on rows of <= 176 tokens qualify, one weight of 512 no longer does
    edge = oracle.synth_queries(80, 4, val_law=synth.VAL_DYADIC)
    got = _search(idx, edge, 100, blocked_postings=1)
    assert got[2].last_packed_tiles == 1
    edge[1, np.nonzero(edge[1])[0][0]] = 512.0
    ref = _search(idx, edge, 100, blocked_postings=0)
    got = _search(idx, edge, 100, blocked_postings=1)
    assert got[2].last_packed_tiles == 0 and (got[0] == ref[0]).all() and (got[1] == ref[1]).all()


def test_prepare_builds_the_postings_copy_ahead_of_the_first_search():
    """vs_index_prepare (the facade calls it from move_to_device / load_index): the copy exists before any search, and
    vs_index_info_t.postings_state says why an index has none."""
    idx = DeviceIndex.synthetic(0, 0, 20000, V, 768, 0, 0, nat.VS_F32)
    assert idx.info().postings_state == 0 and idx.info().aux_bytes == 0
    idx.prepare()
    info = idx.info()
    assert info.postings_state == 1 and info.aux_bytes > 0
    ids, sc, info = _search(idx, oracle.synth_queries(1, 3), 10)
    assert info.last_path == 3
    small = DeviceIndex.synthetic(0, 0, 1000, V, 768, 0, 0, nat.VS_F32)
    small.prepare()
    assert small.info().postings_state == 4 and small.info().aux_bytes == 0
    off = DeviceIndex.synthetic(0, 0, 20000, V, 768, 0, 0, nat.VS_F32)
    off.set_option("blocked_postings", 0)
    off.prepare()
    assert off.info().postings_state == 4


def test_large_k_leaves_the_filter():
    """k + margin beyond the candidate buffers: 'search after' passes of the CSR scan (lossy records) or the fp64 walk."""
    ip, ix, d = oracle.synth_csr(0, 0, 4000)
    q = oracle.synth_queries(1, 5)
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    ref = _search(idx, q, 1500, blocked_postings=0)
    for quant in (-1, 0):
        got = _search(idx, q, 1500, blocked_postings=1, postings_quant=quant)
        assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all()
        assert got[2].queries_per_pass == 8


def test_baseline_sized_index_on_one_gpu():
    """BASELINE.json's index: 21 015 324 docs x 768 nnz, V = 29 523, fp32 -- the bench's configuration (lossy filter copy, 8 lanes
    per list, few chunks) against the dense score matrix of the same index and, bit for bit, against the CSR scan."""
    n = 21_015_324
    idx = DeviceIndex.synthetic(0, 0, n, V, 768, 0, 0, nat.VS_F32)
    q = oracle.synth_queries(1, 4)
    ids, sc, info = _search(idx, q, 100, blocked_postings=-1)
    assert info.last_path == 3 and info.last_fallbacks == 0 and info.aux_bytes > 60e9
    ref_ids, ref_sc, info = _search(idx, q, 100, blocked_postings=0)
    assert info.last_path == 1
    assert (ids == ref_ids).all() and (sc == ref_sc).all()
    allsc = idx.scores(q)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    del allsc
    # the bench's batch shape on the same index: 1024 queries, spot-checked against the CSR scan on a slice
    qb = oracle.synth_queries(2, 1024)
    idx.set_option("blocked_postings", -1)
    ids_b, sc_b = idx.search(qb, 100)
    info = idx.info()
    assert info.last_path == 3 and info.last_fallbacks == 0
    ref_ids, ref_sc, _ = _search(idx, qb[500:516], 100, blocked_postings=0)
    assert (np.asarray(ids_b)[500:516] == ref_ids).all() and (np.asarray(sc_b)[500:516] == ref_sc).all()
    # ... and against the CPU oracle at full size (every 21 M-doc check compared HIP with HIP): the returned rows
    # re-scored, 48 x 8192 sampled rows scored -- for the 4-query search and for 8 queries of the 1024-query batch
    _oracle_pin(0, n, q, ids, sc, 100)
    _oracle_pin(0, n, qb[500:508], np.asarray(ids_b)[500:508], np.asarray(sc_b)[500:508], 100, windows=16)


def test_skewed_columns_device_generator_and_search():
    """KIND_SKEW (column popularity ~ 1 / rank, SURVEY 8(d) C3's secondary run): the device generator equals its host twins row for
    row, and the filter search on it -- long lists for popular columns, saturated head -- stays bit-identical to the CSR scan."""
    n = 9000
    idx = DeviceIndex.synthetic(0, 100, n, V, 768, synth.KIND_SKEW, 0, nat.VS_F32)
    ip, ix, d = idx.export_csr()
    o_ip, o_ix, o_d = oracle.synth_csr(0, 100, n, V, 768, synth.KIND_SKEW)
    assert (ip == o_ip).all() and (ix == o_ix).all() and (d == o_d).all()
    q = oracle.synth_queries(1, 21, kind=synth.KIND_SKEW)
    ids, sc = _modes(idx, q, 100, want_quant=True)
    _, _, allsc = oracle.csr_search(o_ip, o_ix, o_d, V, q, 100, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)


@pytest.mark.parametrize("gemm", [0, 1], ids=["in-walk", "pre-pass"])
@pytest.mark.parametrize("store", [nat.VS_F32, nat.VS_F16], ids=["fp32", "fp16"])
def test_dense_head_strips_keep_the_results(store, gemm):
    """Option postings_head: the columns present in >= 1/N of the documents move from posting lists to dense fp16 strips scored on
    the matrix cores -- inside the walk, tile by tile (postings_head_gemm = 0: auto N = 4, at most 512 columns), or by the head
    pre-pass (bp_head.h, the default: auto N = 8, at most 1024; an index of 1 M documents and more: 16 / 1536).  Whatever N, the results are the CSR scan's, bit for bit; more
    qualifying columns than the cap keep the most frequent."""
    n = 9000
    cap, auto_n = (1024, 8) if gemm else (512, 4)
    idx = DeviceIndex.synthetic(0, 100, n, V, 768, synth.KIND_SKEW, 0, store)
    idx.set_option("postings_head_gemm", gemm)
    ip, ix, d = idx.export_csr()
    df = np.bincount(ix, minlength=V)
    q = oracle.synth_queries(1, 19, kind=synth.KIND_SKEW)
    heads_only = np.zeros((2, V), np.float32)                      # queries that touch head columns only / one ordinary column only
    heads_only[0, np.nonzero(df >= n // 2)[0][:40]] = np.linspace(0.5, 2.0, 40, dtype=np.float32)
    heads_only[1, np.nonzero((df > 0) & (df < n // 100))[0][:3]] = 1.25
    q = np.concatenate([q, heads_only])
    ref_ids, ref_sc, info = _search(idx, q, 100, blocked_postings=0)
    assert info.last_path == 1
    seen = {}
    for head in (0, 2, -1, 8, 64):
        ids, sc, info = _search(idx, q, 100, blocked_postings=1, postings_head=head)
        assert info.last_path == 3, head
        if head == 0 and store == nat.VS_F32:
            # no strips: the skewed corpus goes through the quad walk -- lists of up to a whole block (33 chained chunks), more links
            # than a wave's list holds (the segment mode of bp_quad_topk)
            assert info.postings_walk == 4
        assert (ids == ref_ids).all() and (sc == ref_sc).all(), f"postings_head={head}: differs from the CSR scan"
        seen[head] = info.head_columns
    assert seen[0] == 0
    assert seen[2] == int((df >= -(-n // 2)).sum()) and seen[-1] == min(cap, int((df >= -(-n // auto_n)).sum()))
    lo, mid = (seen[8], seen[-1]) if gemm else (seen[-1], seen[8])                      # (auto: 1/16 with the pre-pass, 1/4 inside the walk)
    assert seen[2] < lo <= mid <= seen[64] <= cap and seen[64] > cap - 32                # (ties at the raised threshold can leave a few below the cap)
    if gemm:
        # the pre-pass in several passes over the batch's 3 tiles (2 + 1), and one tile at a time
        for tiles in (2, 1):
            ids, sc, info = _search(idx, q, 100, blocked_postings=1, postings_head=-1, postings_head_tiles=tiles)
            assert info.last_path == 3 and (ids == ref_ids).all() and (sc == ref_sc).all(), f"{tiles} tiles per pass"
        idx.set_option("postings_head_tiles", 0)
        # both kernels of the pre-pass (the LDS ring is auto's from 32 tiles a pass on: here it is forced onto 3 tiles, 2 + 1 and single ones)
        for product in (1, 0):
            for tiles in (0, 2, 1):
                ids, sc, info = _search(idx, q, 100, blocked_postings=1, postings_head=-1, postings_head_product=product, postings_head_tiles=tiles)
                assert info.last_path == 3 and (ids == ref_ids).all() and (sc == ref_sc).all(), f"head product {product}, {tiles} tiles per pass"
        idx.set_option("postings_head_tiles", 0)
        idx.set_option("postings_head_product", -1)
    _, _, allsc = oracle.csr_search(ip, ix, d.astype(np.float32), V, q, 100, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ref_ids, ref_sc, rtol=RTOL)
    # forced exact pass and the fp64 walk on the same index (the latter rebuilds the copy without strips)
    for opts in (dict(postings_force_fallback=1), dict(postings_filter=0)):
        ids, sc, info = _search(idx, q, 100, blocked_postings=1, postings_head=-1, **opts)
        assert (ids == ref_ids).all() and (sc == ref_sc).all(), opts
        idx.set_option("postings_force_fallback", 0)
    assert info.last_path == 2 and info.head_columns == 0


def test_signed_values_get_no_head_strips():
    """The strips hold fp16 copies and the proof for them needs non-negative values: a signed index keeps every column in lists."""
    n = 6000
    ip, ix, d = oracle.synth_csr(0, 0, n, V, 768, synth.KIND_SKEW)
    d = d * np.where(np.random.default_rng(3).random(len(d)) < 0.25, -1.0, 1.0).astype(np.float32)
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    q = oracle.synth_queries(2, 11, kind=synth.KIND_SKEW)
    ref_ids, ref_sc, _ = _search(idx, q, 50, blocked_postings=0)
    ids, sc, info = _search(idx, q, 50, blocked_postings=1)
    assert info.last_path == 3 and info.head_columns == 0
    assert (ids == ref_ids).all() and (sc == ref_sc).all()


@pytest.mark.parametrize("scale,heads", [(1.0 / 1024, False), (1.0 / 16, True), (900.0, True)], ids=["tiny", "small", "large"])
def test_head_strips_across_value_ranges(scale, heads):
    """The dense part runs on fp16 operands: weights are rescaled by powers of two from the index's largest value, so value ranges
    far from 1 keep the proof (and the results); an index whose values sit near the fp16 subnormals keeps every column in lists."""
    n = 6000
    ip, ix, d = oracle.synth_csr(0, 0, n, V, 768, synth.KIND_SKEW)
    d = (d * np.float32(scale)).astype(np.float32)
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    q = oracle.synth_queries(2, 13, kind=synth.KIND_SKEW) * np.float32(3.0)
    ref_ids, ref_sc, _ = _search(idx, q, 100, blocked_postings=0)
    ids, sc, info = _search(idx, q, 100, blocked_postings=1)
    assert info.last_path == 3 and (info.head_columns > 0) == heads
    assert info.last_fallbacks == 0
    assert (ids == ref_ids).all() and (sc == ref_sc).all()


@pytest.mark.parametrize("store", [nat.VS_F32, nat.VS_F16, nat.VS_NONE], ids=["fp32", "fp16", "binary"])
def test_record_layout_options_keep_the_results(store):
    """postings_align (lists start on whole 128-byte lines), postings_lanes and postings_rows only move postings around: every
    combination returns the CSR scan's ids and scores bit for bit.  Rows 700 .. 720 get a column in common so that its list in
    one block is longer than one round of 8 lanes x 8 postings (the second record a lane takes in the same load round)."""
    n = 9000
    if store == nat.VS_NONE:
        idx = DeviceIndex.synthetic(0, 0, 70000, V, 86, synth.KIND_BOT, 0, store)
        q = oracle.synth_queries(1, 11, V, 776, synth.VAL_DYADIC)
    else:
        ip, ix, d = oracle.synth_csr(0, 0, n)
        ix = ix.copy()
        for r in range(600, 900):                                    # 300 documents of one block share column 77
            if 77 not in ix[ip[r]:ip[r + 1]]:
                ix[ip[r]] = 77
                ix[ip[r]:ip[r + 1]].sort()
        idx = DeviceIndex.from_csr(ip, ix, d, V, store_dtype=store)
        q = oracle.synth_queries(1, 11)
        q[:, 77] = np.float32(1.5)
    ref_ids, ref_sc, info = _search(idx, q, 100, blocked_postings=0)
    assert info.last_path == 1
    for align, lanes, rows in [(0, 0, 0), (1, 0, 0), (1, 4, 0), (0, 4, 0), (1, 8, 1024), (0, 8, 256)]:
        ids, sc, info = _search(idx, q, 100, blocked_postings=1, postings_align=align, postings_lanes=lanes, postings_rows=rows)
        assert info.last_path == 3, (align, lanes, rows)
        assert (ids == ref_ids).all() and (sc == ref_sc).all(), f"align={align} lanes={lanes} rows={rows}: differs from the CSR scan"


def test_baseline_sized_bag_of_token_index():
    """C5 at full size: 21 015 324 docs, ~86 binary non-zeros each, dyadic query weights -- the postings walk returns the oracle's
    integer-exact scores: bit-equal to the CSR scan, canonical ids, a valid top-k of the score matrix."""
    n = 21_015_324
    idx = DeviceIndex.synthetic(0, 0, n, V, 86, synth.KIND_BOT, 0, nat.VS_NONE)
    q = oracle.synth_queries(1, 8, V, 776, synth.VAL_DYADIC)
    ids, sc, info = _search(idx, q, 100, blocked_postings=-1)
    assert info.last_path == 3 and info.last_fallbacks == 0
    ref_ids, ref_sc, info = _search(idx, q, 100, blocked_postings=0)
    assert info.last_path == 1
    assert (ids == ref_ids).all() and (sc == ref_sc).all()
    allsc = idx.scores(q[:4])
    compare.check_topk_valid(allsc, ids[:4], sc[:4], rtol=RTOL, exact=True, canonical=True)
    del allsc
    # the CPU oracle at full size: returned rows re-scored (bit-exact: integer-scaled sums), sampled rows scored
    _oracle_pin(0, n, q[:4], ids[:4], sc[:4], 100, kind=synth.KIND_BOT, nnz=86, windows=32, window_rows=65536, binary=True, exact=True)


def test_baseline_sized_skewed_index_with_head_strips():
    """C3's secondary column law at C4's size: 21 015 324 docs x 768 nnz with Zipf column popularity; 1536 head columns become dense
    strips, their part of the sums comes from the head pre-pass (bp_head.h).  Bit-equal to the CSR scan on a batch slice, no query falls back."""
    n = 21_015_324
    idx = DeviceIndex.synthetic(0, 0, n, V, 768, synth.KIND_SKEW, 0, nat.VS_F32)
    q = oracle.synth_queries(1, 64, kind=synth.KIND_SKEW)
    ids, sc, info = _search(idx, q, 100, blocked_postings=-1)
    assert info.last_path == 3 and info.last_fallbacks == 0 and info.head_columns > 1024          # (the wide head of a large index)
    ref_ids, ref_sc, info = _search(idx, q[20:28], 100, blocked_postings=0)
    assert info.last_path == 1
    assert (ids[20:28] == ref_ids).all() and (sc[20:28] == ref_sc).all()
    _oracle_pin(0, n, q[20:24], ids[20:24], sc[20:24], 100, kind=synth.KIND_SKEW, windows=24)
    # the bench's batch shape (1024 queries = 128 tiles: two passes of the head pre-pass over its 43 GB scratch), a slice against the CSR scan
    qb = oracle.synth_queries(2, 1024, kind=synth.KIND_SKEW)
    ids_b, sc_b, info = _search(idx, qb, 100, blocked_postings=-1)
    assert info.last_path == 3 and info.last_fallbacks == 0 and info.head_columns > 512
    ref_ids, ref_sc, _ = _search(idx, qb[700:708], 100, blocked_postings=0)
    assert (ids_b[700:708] == ref_ids).all() and (sc_b[700:708] == ref_sc).all()


def test_c3_shape_one_million_docs_1024_queries():
    """BASELINE.json configs[2] at its exact shape: 1 M synthetic docs x 768 nnz sparse CSR, batch 1024, k = 100 -- the filter path against
    the CSR scan on a slice, bit for bit, and pinned to the CPU oracle."""
    n = 1_000_000
    idx = DeviceIndex.synthetic(0, 0, n, V, 768, 0, 0, nat.VS_F32)
    q = oracle.synth_queries(1, 1024)
    ids, sc, info = _search(idx, q, 100, blocked_postings=-1)
    assert info.last_path == 3 and info.last_fallbacks == 0 and info.postings_walk == 4
    ref_ids, ref_sc, info = _search(idx, q[300:316], 100, blocked_postings=0)
    assert info.last_path == 1
    assert (ids[300:316] == ref_ids).all() and (sc[300:316] == ref_sc).all()
    _oracle_pin(0, n, q[300:304], ids[300:304], sc[300:304], 100, windows=16)


def test_head_strips_edge_cases_empty_queries_and_append():
    """Dense strips with queries that have no entries at all / entries on head columns only / a single entry, in one ragged batch;
    appending rows drops the strips with the rest of the copy and the next search rebuilds them."""
    n = 12000
    ip, ix, d = oracle.synth_csr(0, 0, n, V, 768, synth.KIND_SKEW)
    cut = int(ip[9000])
    idx = DeviceIndex.reserved(n, len(ix), V, nat.VS_F32)
    idx.append_csr(ip[:9001], ix[:cut], d[:cut])
    df = np.bincount(ix[:cut], minlength=V)
    q = oracle.synth_queries(3, 12, kind=synth.KIND_SKEW)
    q[1] = 0.0                                                        # no entries
    q[5] = 0.0
    q[5, np.nonzero(df >= 9000 // 2)[0][:9]] = 2.0                   # head columns only
    q[7] = 0.0
    q[7, int(np.nonzero((df > 3) & (df < 40))[0][0])] = 0.75          # one ordinary column
    q[11] = 0.0                                                       # the batch ends on an empty query
    for stage in ("9000 rows", "12000 rows"):
        ref_ids, ref_sc, info = _search(idx, q, 50, blocked_postings=0)
        ids, sc, info = _search(idx, q, 50, blocked_postings=1)
        assert info.last_path == 3 and info.head_columns > 100, stage
        assert (ids == ref_ids).all() and (sc == ref_sc).all(), stage
        if stage == "9000 rows":
            idx.append_csr(ip[9000:] - ip[9000], ix[cut:], d[cut:])
            assert idx.info().head_columns == 0                       # the copy (lists and strips) is gone until the next search


def test_sparse_rows_take_the_record_copy_not_quad_chunks():
    """VERDICT r4 item 5 / ADVICE r4: a valued index of 128-nnz documents.  Quad chunks would cost n_blocks x V x 256 bytes (5 x the CSR)
    for 86 %-empty chunks; the auto policy must pick the record copy (postings_walk 0), stay under 3 x the CSR bytes, take the filter
    path and match the oracle; forcing quad chunks (postings_walk = 4) must still be bit-identical."""
    n, nnz = 40_000, 128
    ip, ix, d = oracle.synth_csr(0, 0, n, V, nnz)
    idx = DeviceIndex.from_csr(ip, ix, d, V, store_dtype=nat.VS_F32)
    q = oracle.synth_queries(1, 16)
    ids, sc, info = _search(idx, q, 100, blocked_postings=1)
    assert info.last_path == 3 and info.postings_walk == 0, (info.last_path, info.postings_walk)
    csr_bytes = info.device_bytes
    assert 0 < info.aux_bytes < 3 * csr_bytes, (info.aux_bytes, csr_bytes)
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q, 100, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL)
    ids4, sc4, info4 = _search(idx, q, 100, postings_walk=4)
    assert info4.last_path == 3 and info4.postings_walk == 4
    assert (ids4 == ids).all() and (sc4 == sc).all()
    assert info4.aux_bytes > info.aux_bytes                    # (what the gate avoids)
    idx.close()
