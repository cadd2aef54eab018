"""Tie-aware comparison of top-k results (TEST INFRASTRUCTURE).

``torch.topk`` leaves the order of equal scores unspecified (SURVEY.md §7 "Tie semantics":
``[1,3,3,2,3,3,0,3].topk(3)`` -> ids ``[4,7,5]``) and fp32 sums depend on accumulation order
(max rel err 1.3e-7 observed), so "identical ids" is defined as:

  * the score sequences agree (exactly on the exactly-summable binary x dyadic path, else within
    ``rtol`` = 1e-4 relative, the tolerance BASELINE.json's north_star states);
  * within every run of equal (or, on fp32 paths, nearly equal: gap < ``tie_rtol``) scores the id
    *sets* agree; the run that is cut by rank k may hold any members of that run.
"""
from __future__ import annotations

import numpy as np


def recall_at_k(ids_a, ids_b):
    ids_a, ids_b = np.asarray(ids_a), np.asarray(ids_b)
    k = ids_a.shape[1]
    return float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(ids_a, ids_b)]))


def _runs(scores, tie_rtol):
    """Split one descending score row into runs of (nearly) equal scores -> list of (start, stop)."""
    k = len(scores)
    if k == 0:
        return []
    s = scores.astype(np.float64)
    gap = s[:-1] - s[1:]
    brk = gap > tie_rtol * np.maximum(np.abs(s[:-1]), np.abs(s[1:]))
    cuts = [0] + (np.nonzero(brk)[0] + 1).tolist() + [k]
    return list(zip(cuts[:-1], cuts[1:]))


def compare_topk(ids_ref, scores_ref, ids_got, scores_got, rtol=1e-4, exact=False, tie_rtol=1e-6):
    """Raise AssertionError with a precise message unless ``got`` matches ``ref`` (see module doc)."""
    ids_ref, ids_got = np.asarray(ids_ref).astype(np.int64), np.asarray(ids_got).astype(np.int64)
    scores_ref, scores_got = np.asarray(scores_ref, np.float32), np.asarray(scores_got, np.float32)
    assert ids_ref.shape == ids_got.shape == scores_ref.shape == scores_got.shape, \
        (ids_ref.shape, ids_got.shape, scores_ref.shape, scores_got.shape)
    if exact:
        bad = scores_ref != scores_got
        assert not bad.any(), f"scores differ bit-wise at {np.argwhere(bad)[:5].tolist()}: " \
                              f"{scores_ref[bad][:5]} vs {scores_got[bad][:5]}"
    else:
        denom = np.maximum(np.abs(scores_ref), 1e-30)
        rel = np.abs(scores_ref.astype(np.float64) - scores_got) / denom
        assert rel.max(initial=0) <= rtol, f"max rel score err {rel.max():.3e} > {rtol}"
    k = ids_ref.shape[1]
    for b in range(ids_ref.shape[0]):
        d = np.diff(scores_got[b].astype(np.float64))
        assert (d <= 0).all(), f"row {b}: scores not descending"
        for lo, hi in _runs(scores_ref[b], 0.0 if exact else tie_rtol):
            if hi == k:
                continue                      # run cut by rank k: any members of the run are valid
            a, g = set(ids_ref[b, lo:hi].tolist()), set(ids_got[b, lo:hi].tolist())
            assert a == g, f"row {b} ranks [{lo},{hi}): id sets differ: ref-only {sorted(a - g)[:5]}, got-only {sorted(g - a)[:5]}"
        assert len(set(ids_got[b].tolist())) == k, f"row {b}: duplicate ids"


def check_topk_valid(all_scores, ids, scores, rtol=1e-4, exact=False, canonical=False):
    """Check (ids, scores) is a valid top-k of the dense score matrix ``all_scores`` [B,N]
    (the matrix the reference materialises at index.py:91).  Stronger than compare_topk: also
    verifies the members of the run cut by rank k.  ``canonical``: additionally require the
    library's documented order (score desc, id asc) with lowest ids winning boundary ties.
    """
    all_scores = np.asarray(all_scores, np.float32)
    ids = np.asarray(ids).astype(np.int64)
    scores = np.asarray(scores, np.float32)
    B, k = ids.shape
    for b in range(B):
        row = all_scores[b].astype(np.float64)
        assert len(set(ids[b].tolist())) == k, f"row {b}: duplicate ids"
        assert ids[b].min() >= 0 and ids[b].max() < row.shape[0], f"row {b}: id out of range"
        true = row[ids[b]]
        if exact:
            assert (true.astype(np.float32) == scores[b]).all(), f"row {b}: returned scores != true scores (bit-wise)"
        else:
            rel = np.abs(true - scores[b]) / np.maximum(np.abs(true), 1e-30)
            assert rel.max() <= rtol, f"row {b}: max rel err {rel.max():.3e}"
        assert (np.diff(scores[b].astype(np.float64)) <= 0).all(), f"row {b}: not descending"
        kth = np.partition(row, -k)[-k]                     # true k-th best score
        tol = 0.0 if exact else rtol * max(abs(kth), 1e-30)
        must = np.nonzero(row > kth + tol)[0]               # strictly better than the k-th: mandatory
        missing = set(must.tolist()) - set(ids[b].tolist())
        assert not missing, f"row {b}: missed {len(missing)} docs scoring above the k-th best, e.g. {sorted(missing)[:5]}"
        assert true.min() >= kth - tol, f"row {b}: returned a doc below the k-th best ({true.min()} < {kth})"
        if canonical:
            order = np.lexsort((np.arange(row.shape[0]), -row))[:k]
            if exact:
                assert (order == ids[b]).all(), f"row {b}: not in canonical (score desc, id asc) order"
            else:
                srt = np.lexsort((ids[b], -scores[b].astype(np.float64)))
                assert (srt == np.arange(k)).all(), f"row {b}: returned list not in (score desc, id asc) order"
