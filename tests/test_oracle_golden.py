"""Pin the CPU oracle (oracle/) against golden vectors captured from the reference itself
(tools/gen_golden.py, run in the build container with /root/reference imported).  CPU-only.

Every test states the reference call site the fixture came from.
"""
import numpy as np
import pytest

import oracle
from oracle import compare
from conftest import V, VOCAB, SHIFT
from vsearch_amd import synth


# --- synthetic generator twins -------------------------------------------------------------------
@pytest.mark.parametrize("kind,nnz,law", [(synth.KIND_VDR, 768, synth.VAL_GRID), (synth.KIND_BOT, 86, synth.VAL_ONE),
                                          (synth.KIND_VDR, 776, synth.VAL_DYADIC)])
def test_synth_numpy_equals_c(kind, nnz, law):
    a = synth.synth_csr(7, 1000, 64, V, nnz, kind, law)
    b = oracle.synth_csr(7, 1000, 64, V, nnz, kind, law)
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and (x == y).all()


def test_synth_rows_are_canonical_and_stateless():
    ip, ix, d = oracle.synth_csr(0, 0, 200)
    assert (np.diff(ip) == 768).all()
    for r in range(200):
        row = ix[ip[r]:ip[r + 1]]
        assert (np.diff(row) > 0).all() and row.min() >= 0 and row.max() < V
    ip2, ix2, d2 = oracle.synth_csr(0, 150, 50)           # rows are a pure function of (seed, row id)
    assert (ix2 == ix[ip[150]:]).all() and (d2 == d[ip[150]:]).all()


# --- src/ir/utils/sparse.py ----------------------------------------------------------------------
def test_elu1p(golden):
    g = golden("sparse_utils")                                           # sparse.py:6
    # elu(x)+1 cancels for x << 0 (expm1 -> -1): the result carries the absolute error of one ulp at 1.0
    np.testing.assert_allclose(oracle.elu1p(g["elu_in"]), g["elu_out"], rtol=2e-7, atol=1.2e-7)


@pytest.mark.parametrize("k", [1, 100, 768])
def test_topk_mask(golden, k):
    g = golden("sparse_utils")                                           # sparse.py:8-14
    x = synth.dense_tiefree(int(g["x_seed"]), (int(g["x_rows"]), V))
    ref = np.unpackbits(g[f"mask_k{k}"], axis=1)[:, :V].astype(bool)
    assert (oracle.topk_mask(x, k) == ref).all()


def test_topk_sparsify(golden):
    g = golden("sparse_utils")                                           # sparse.py:16-19
    x = synth.dense_tiefree(int(g["x_seed"]), (int(g["x_rows"]), V))
    got = x * oracle.topk_mask(x, 768)
    r, c = np.nonzero(got)
    assert (c.reshape(-1, 768) == g["sparsify_cols"]).all()
    assert (got[r, c].reshape(-1, 768) == g["sparsify_vals"]).all()


@pytest.mark.parametrize("batch", [0, 1])
@pytest.mark.parametrize("norm", [False, True])
def test_bow_mask(golden, batch, norm):
    g = golden("bow_mask")                                               # sparse.py:21-29
    ids = g[f"b{batch}_ids"]
    got = oracle.bow_mask(ids, VOCAB, SHIFT, norm)
    tag = f"b{batch}_{'norm' if norm else 'raw'}"
    r, c = np.nonzero(got)
    assert (r == g[f"{tag}_rows"]).all() and (c == g[f"{tag}_cols"]).all()
    np.testing.assert_allclose(got[r, c], g[f"{tag}_vals"], rtol=1e-6)
    assert (oracle.bow_mask(ids, VOCAB, 0).sum(1) == g[f"b{batch}_noshift_nnz"]).all()


# --- src/ir/encoder/vdr.py -----------------------------------------------------------------------
def test_encoder_head(golden):
    """vdr.py:71-75: the oracle pools the [B,L,V] logits; LN + projection (MFMA GEMM territory)
    are recomputed here in float64 so that only the head's tail is under test."""
    g = golden("encoder_head")
    B, L, H, vocab, shift = g["shape"].tolist()
    s = g["seeds"].tolist()
    hidden = synth.dense_uniform(s[0], (B, L, H), -2.0, 2.0).astype(np.float64)
    W = synth.dense_uniform(s[1], (vocab, H), -0.08, 0.08).astype(np.float64)
    ln_w = synth.dense_uniform(s[2], (H,), 0.5, 1.5).astype(np.float64)
    ln_b = synth.dense_uniform(s[3], (H,), -0.1, 0.1).astype(np.float64)
    mu = hidden.mean(-1, keepdims=True)
    var = hidden.var(-1, keepdims=True)
    h_ln = (hidden - mu) / np.sqrt(var + 1e-5) * ln_w + ln_b
    logits = (h_ln @ W[shift:].T).astype(np.float32)
    emb = oracle.head_pool(logits)
    np.testing.assert_allclose(emb, g["emb"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logits.max(1), g["logits_max"], rtol=1e-4, atol=2e-5)
    # elu1p commutes with max (monotone): pooled-then-activated == activated-then-pooled
    assert (oracle.elu1p(logits.max(1)) == emb).all()
    nrm = emb / np.maximum(np.sqrt((emb.astype(np.float64) ** 2).sum(1, keepdims=True)), 1e-12)
    np.testing.assert_allclose(nrm, g["emb_norm"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("name,kw", [
    ("top768_lex", dict(topk=768, activate_lexical=True)),
    ("top768_nolex", dict(topk=768, activate_lexical=False)),
    ("top0_lex", dict(topk=0, activate_lexical=True)),
    ("bow", dict(bow=True)),
    ("top16_lex_bs4", dict(topk=16, activate_lexical=True)),
])
def test_embed_mask(golden, name, kw):
    g = golden("embed_mask")                                             # vdr.py:152-169
    ids = g["ids"]
    dense = synth.dense_tiefree(int(g["dense_seed"]), (ids.shape[0], V), 0.05, 6.0)
    got = oracle.embed_mask(dense, ids, VOCAB, SHIFT, **kw)
    r, c = np.nonzero(got)
    assert (r == g[f"{name}_rows"]).all() and (c == g[f"{name}_cols"]).all()
    assert (got[r, c] == g[f"{name}_vals"]).all()


# --- src/ir/retriever/index.py:88-94 -------------------------------------------------------------
@pytest.mark.parametrize("name,ks", [("search_sparse_n2000", (1, 100, 2000)), ("search_sparse_n20000", (100,))])
def test_sparse_search(golden, name, ks):
    g = golden(name)
    n, b = int(g["n"]), int(g["b"])
    ip, ix, d = oracle.synth_csr(int(g["index_seed"]), 0, n)
    q = oracle.synth_queries(int(g["query_seed"]), b)
    for k in ks:
        ids, sc, allsc = oracle.csr_search(ip, ix, d, V, q, k, return_all=True)
        compare.compare_topk(g[f"ids_k{k}"], g[f"scores_k{k}"], ids, sc, rtol=1e-4)
        compare.check_topk_valid(allsc, g[f"ids_k{k}"], g[f"scores_k{k}"], rtol=1e-4)   # reference output is a valid top-k of oracle scores
        compare.check_topk_valid(allsc, ids, sc, exact=True, canonical=True)
        assert compare.recall_at_k(g[f"ids_k{k}"], ids) >= 0.999


def test_sparse_search_torch_ref_port(golden):
    """The restated three torch calls (oracle/torch_ref.py) reproduce the golden bit-for-bit on
    this container (same torch build) -- the CPU-baseline port is the reference's arithmetic."""
    import torch
    from oracle import torch_ref
    g = golden("search_sparse_n2000")
    ip, ix, d = oracle.synth_csr(0, 0, 2000)
    q = oracle.synth_queries(1, 8)
    ids, sc = torch_ref.search(torch_ref.make_csr(ip, ix, d, (2000, V)), torch.from_numpy(q), 100)
    compare.compare_topk(g["ids_k100"], g["scores_k100"], ids.numpy(), sc.numpy(), rtol=1e-6)


def test_dense_search(golden):
    g = golden("search_dense")
    n, b = int(g["n"]), int(g["b"])
    ip, ix, d = oracle.synth_csr(0, 0, n)
    dense = np.zeros((n, V), np.float32)
    dense[np.repeat(np.arange(n), 768), ix] = d
    q = oracle.synth_queries(1, b)
    for k in (1, 100):
        ids, sc = oracle.dense_search(dense, q, k)
        compare.compare_topk(g[f"ids_k{k}"], g[f"scores_k{k}"], ids, sc, rtol=1e-4)
    n2, b2, s1, s2 = g["full_shape"].tolist()
    # 29 523-term fp32 dot products: a left-to-right fp32 sum is off by ~1e-5 rel, the reference's blocked GEMM
    # by ~1e-6; compare against the correctly-rounded (fp64-accumulated) oracle with a matching near-tie window
    ids, sc = oracle.dense_search(synth.dense_uniform(s1, (n2, V), 0.0, 1.0), synth.dense_uniform(s2, (b2, V), 0.0, 1.0), 50, acc64=True)
    compare.compare_topk(g["full_ids_k50"], g["full_scores_k50"], ids, sc, rtol=1e-4, tie_rtol=3e-6)
    assert bool(g["k_gt_n_raises"])
    with pytest.raises(RuntimeError):
        oracle.dense_search(dense[:10], q, 11)


@pytest.mark.parametrize("tag,exact", [("f32", False), ("dyadic", True)])
def test_bot_search(golden, tag, exact):
    g = golden("search_bot")                                             # BoTIndex inherits search (index.py:205-218)
    n, b = int(g["n"]), int(g["b"])
    ip, ix, _ = oracle.synth_csr(int(g["index_seed"]), 0, n, V, int(g["nnz"]), synth.KIND_BOT)
    seed = int(g["query_seeds"][1 if exact else 0])
    q = oracle.synth_queries(seed, b, val_law=synth.VAL_DYADIC if exact else synth.VAL_GRID)
    for k in (10, 100):
        ids, sc, allsc = oracle.csr_search(ip, ix, None, V, q, k, return_all=True)
        compare.compare_topk(g[f"{tag}_ids_k{k}"], g[f"{tag}_scores_k{k}"], ids, sc, rtol=1e-4, exact=exact)
        compare.check_topk_valid(allsc, g[f"{tag}_ids_k{k}"], g[f"{tag}_scores_k{k}"], rtol=1e-4, exact=exact)


# --- src/ir/retriever/retriever.py:208-253 -------------------------------------------------------
def _tokenize(texts, max_len=128):
    out = []
    for t in texts:
        ids = [int(x) for x in str(t).split()]
        if len(ids) > max_len:
            ids = ids[:max_len - 1] + [102]
        out.append(ids)
    return out


@pytest.mark.parametrize("tag,kw,max_len", [("full", {}, 128), ("max16", {"max_token": 16}, 128), ("len32", {}, 32), ("fp32", {}, 128)])
def test_bot_build(golden, tag, kw, max_len):
    g = golden("bot_build")
    toks = _tokenize(g["texts"].tolist(), max_len)
    ip, ix = oracle.bot_build(toks, VOCAB, SHIFT, **kw)
    assert (ip == g[f"{tag}_indptr"]).all() and (ix == g[f"{tag}_indices"]).all()
    assert g[f"{tag}_shape"].tolist() == [len(toks), V]
    assert str(g[f"{tag}_dtype"]) == ("torch.float32" if tag == "fp32" else "torch.float16")


# --- Retriever.retrieve (retriever.py:107-148) ---------------------------------------------------
def test_retrieve_and_rerank(golden):
    g = golden("retrieve")
    n, b, k = int(g["n"]), int(g["b"]), int(g["k"])
    s_bot, s_par, s_q = g["seeds"].tolist()
    ip, ix, _ = oracle.synth_csr(s_bot, 0, n, V, 86, synth.KIND_BOT)
    q = oracle.synth_queries(s_q, b)
    ids, sc = oracle.csr_search(ip, ix, None, V, q, k)
    compare.compare_topk(g["ids"], g["scores"], ids, sc, rtol=1e-4)
    # rerank (retriever.py:137-147): re-embed the k hits, bmm with q, topk, gather
    ip2, ix2, d2 = oracle.synth_csr(s_par, 0, n)
    rid = np.empty_like(ids)
    rsc = np.empty_like(sc)
    for i in range(b):
        sub = g["ids"][i].astype(np.int64)                 # rerank the reference's own first-stage hits
        rows = [(ix2[ip2[j]:ip2[j + 1]], d2[ip2[j]:ip2[j + 1]]) for j in sub]
        s = np.array([np.float32((q[i, c].astype(np.float32) * v).astype(np.float32).sum(dtype=np.float32)) for c, v in rows], np.float32)
        o = np.lexsort((np.arange(k), -s.astype(np.float64)))
        rid[i], rsc[i] = sub[o], s[o]
    compare.compare_topk(g["rerank_ids"], g["rerank_scores"], rid, rsc, rtol=1e-4)
    ids_s, sc_s = oracle.csr_search(ip2, ix2, d2, V, q, k)
    compare.compare_topk(g["sparse_ids"], g["sparse_scores"], ids_s, sc_s, rtol=1e-4)
    assert bool(g["bad_query_raises"])


# --- SparseIndex.save / load (index.py:163-202) --------------------------------------------------
def test_save_load_manifest(golden):
    g = golden("save_load")
    assert g["manifest_keys"].tolist() == ["_is_array", "data", "format", "indices", "indptr", "shape"]
    assert g["manifest_format"].tolist() in (b"csr", "csr")
    assert g["manifest_shape"].tolist() == [10, V]
    ip, ix, d = oracle.synth_csr(int(g["seed"]), 0, 20)
    assert (g["reload_indptr"] == ip[:11]).all() and (g["reload_indices"] == ix[:ip[10]]).all()
    assert (g["reload_data"] == d[:ip[10]]).all()
    assert g["shift0_shape"].tolist() == [20, V]
    assert (g["shift0_indptr"] == ip).all() and (g["shift0_indices"] == ix).all() and (g["shift0_data"] == d).all()
    # shift=999 drops columns < 999 and renumbers (load_npz(f)[:, shift:], index.py:172)
    keep = ix >= 999
    assert g["shift999_shape"].tolist() == [20, V - 999]
    assert (g["shift999_indices"] == ix[keep] - 999).all() and (g["shift999_data"] == d[keep]).all()
    cnt = np.add.reduceat(keep.astype(np.int64), ip[:-1])
    assert (np.diff(g["shift999_indptr"]) == cnt).all()


def test_skewed_column_law_twins_agree():
    """KIND_SKEW rows: numpy generator == C twin; rows are sorted, duplicate-free, of the requested length; the head ranks sit in
    every row and popularity falls off like 1 / rank."""
    import oracle
    from vsearch_amd import synth
    for n_cols, nnz in [(29523, 768), (29523, 776), (5000, 300), (300, 100)]:
        a = synth.synth_csr(3, 5, 30, n_cols, nnz, synth.KIND_SKEW)
        b = oracle.synth_csr(3, 5, 30, n_cols, nnz, synth.KIND_SKEW)
        assert all((x == y).all() for x, y in zip(a, b))
        assert (np.diff(a[0]) == min(nnz, n_cols)).all()
        for r in range(30):
            c = a[1][a[0][r]:a[0][r + 1]]
            assert (np.diff(c) > 0).all()
    ip, ix, _ = oracle.synth_csr(0, 0, 1500, 29523, 768, synth.KIND_SKEW)
    freq = np.sort(np.bincount(ix, minlength=29523))[::-1]
    assert (freq[:127] == 1500).all()                                   # saturated head
    assert 0.5 < freq[127] / 1500 < 0.8 and 0.25 < freq[255] / 1500 < 0.45 and 0.03 < freq[2047] / 1500 < 0.12 and freq[20000] / 1500 < 0.02
