"""GPU parity tests of the Python facade (vsearch_amd.ir == the reference's src.ir surface):
sparsify helpers, Index / SparseIndex / BoTIndex, Retriever.retrieve / build_index / save / load,
VDREncoder.embed -- against the reference goldens and the CPU oracle.  Run on MI355X.
"""
import os
import types

import numpy as np
import pytest
import torch

import oracle
from oracle import compare
from conftest import V, VOCAB, SHIFT
from vsearch_amd import synth
from vsearch_amd.ir import BoTIndex, Index, IndexType, Retriever, SearchResults, SparseIndex
from vsearch_amd.ir.utils import sparse as sp

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def csr_tensor(ip, ix, d, n_cols=V):
    return torch.sparse_csr_tensor(torch.from_numpy(ip), torch.from_numpy(ix.astype(np.int64)), torch.from_numpy(d), size=(len(ip) - 1, n_cols))


# ---- src/ir/utils/sparse.py ---------------------------------------------------------------------------
def test_elu1p(golden):
    g = golden("sparse_utils")
    out = sp.elu1p(torch.from_numpy(g["elu_in"]))
    np.testing.assert_allclose(out.numpy(), g["elu_out"], rtol=2e-7, atol=1.2e-7)
    assert sp.elu1p(torch.from_numpy(g["elu_in"]).cuda()).is_cuda


@pytest.mark.parametrize("k", [1, 100, 768])
def test_build_topk_mask(golden, k):
    g = golden("sparse_utils")
    x = synth.dense_tiefree(int(g["x_seed"]), (int(g["x_rows"]), V))
    ref = np.unpackbits(g[f"mask_k{k}"], axis=1)[:, :V].astype(bool)
    m = sp.build_topk_mask(torch.from_numpy(x), k=k)
    assert m.dtype == torch.bool and (m.numpy() == ref).all()
    assert (sp.build_topk_mask(x, k=k).numpy() == ref).all()             # numpy input branch (sparse.py:9-10)


def test_topk_mask_ties_and_edges():
    x = torch.tensor([[1., 3., 3., 2., 3., 3., 0., 3.], [5., 5., 5., 5., 5., 5., 5., 5.]])
    m = sp.build_topk_mask(x, k=3)
    assert m.tolist() == [[False, True, True, False, True, False, False, False], [True, True, True] + [False] * 5]
    assert (m.numpy() == oracle.topk_mask(x.numpy(), 3)).all()
    assert sp.build_topk_mask(x, k=8).all() and not sp.build_topk_mask(x, k=0).any()
    with pytest.raises(RuntimeError):
        sp.build_topk_mask(x, k=9)
    neg = torch.tensor([[-1., -3., -0.5, -2.]])
    assert sp.build_topk_mask(neg, k=2).tolist() == [[True, False, True, False]]


def test_topk_sparsify(golden):
    g = golden("sparse_utils")
    x = synth.dense_tiefree(int(g["x_seed"]), (int(g["x_rows"]), V))
    out = sp.topk_sparsify(torch.from_numpy(x).cuda(), 768).cpu().numpy()
    r, c = np.nonzero(out)
    assert (c.reshape(-1, 768) == g["sparsify_cols"]).all() and (out[r, c].reshape(-1, 768) == g["sparsify_vals"]).all()


@pytest.mark.parametrize("batch", [0, 1])
@pytest.mark.parametrize("norm", [False, True])
def test_build_bow_mask(golden, batch, norm):
    g = golden("bow_mask")
    ids = g[f"b{batch}_ids"]
    out = sp.build_bow_mask(torch.from_numpy(ids), vocab_size=VOCAB, shift_num=SHIFT, norm=norm).numpy()
    tag = f"b{batch}_{'norm' if norm else 'raw'}"
    r, c = np.nonzero(out)
    assert out.shape == (ids.shape[0], V) and (r == g[f"{tag}_rows"]).all() and (c == g[f"{tag}_cols"]).all()
    np.testing.assert_allclose(out[r, c], g[f"{tag}_vals"], rtol=1e-6)
    assert (sp.build_bow_mask(torch.from_numpy(ids), VOCAB, 0).sum(1).numpy() == g[f"b{batch}_noshift_nnz"]).all()
    with pytest.raises(RuntimeError):
        sp.build_bow_mask(torch.tensor([[VOCAB]]), VOCAB, SHIFT)


@pytest.mark.parametrize("name,kw", [
    ("top768_lex", dict(topk=768, activate_lexical=True)), ("top768_nolex", dict(topk=768, activate_lexical=False)),
    ("top0_lex", dict(topk=0, activate_lexical=True)), ("bow", dict(topk=0, activate_lexical=True, bow=True)),
    ("top16_lex_bs4", dict(topk=16, activate_lexical=True)),
])
def test_embed_mask_stage(golden, name, kw):
    g = golden("embed_mask")                                             # vdr.py:152-169
    ids = torch.from_numpy(g["ids"])
    emb = torch.from_numpy(synth.dense_tiefree(int(g["dense_seed"]), (ids.shape[0], V), 0.05, 6.0)).cuda()
    sp.apply_embed_mask_(emb, ids, VOCAB, SHIFT, **kw)
    out = emb.cpu().numpy()
    r, c = np.nonzero(out)
    assert (r == g[f"{name}_rows"]).all() and (c == g[f"{name}_cols"]).all() and (out[r, c] == g[f"{name}_vals"]).all()


def test_head_pool_and_dense_to_csr(golden):
    g = golden("encoder_head")
    B, L, H, vocab, shift = g["shape"].tolist()
    s = g["seeds"].tolist()
    hidden = torch.from_numpy(synth.dense_uniform(s[0], (B, L, H), -2.0, 2.0)).cuda()
    W = torch.from_numpy(synth.dense_uniform(s[1], (vocab, H), -0.08, 0.08)).cuda()
    ln = torch.nn.LayerNorm(H).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.from_numpy(synth.dense_uniform(s[2], (H,), 0.5, 1.5)))
        ln.bias.copy_(torch.from_numpy(synth.dense_uniform(s[3], (H,), -0.1, 0.1)))
        logits = ln(hidden) @ W[shift:].t()
    emb = sp.head_pool(logits)
    np.testing.assert_allclose(emb.cpu().numpy(), g["emb"], rtol=1e-4, atol=1e-5)
    assert (emb.cpu().numpy() == oracle.head_pool(logits.cpu().numpy())).all()      # same logits -> bit-equal pooling
    masked = sp.topk_sparsify(emb, 64)
    rp, ci, va = sp.dense_to_csr(masked)
    want = masked.cpu().to_sparse_csr()
    assert (rp.cpu() == want.crow_indices()).all() and (ci.cpu() == want.col_indices()).all() and (va.cpu() == want.values()).all()


# ---- Index / SparseIndex / BoTIndex --------------------------------------------------------------------
@pytest.mark.parametrize("device", ["cpu", "cuda"])
def test_sparse_index_search_golden(golden, device):
    g = golden("search_sparse_n2000")
    ip, ix, d = oracle.synth_csr(0, 0, 2000)
    idx = SparseIndex(device=device)
    idx.vector = csr_tensor(ip, ix, d)
    idx.move_to_device(device)
    q = torch.from_numpy(oracle.synth_queries(1, 8))
    res = idx.search(q, 100)
    assert isinstance(res, SearchResults) and res.ids.dtype == torch.int64 and res.scores.dtype == torch.float32
    assert res.ids.device.type == device and tuple(res.ids.shape) == (8, 100)
    compare.compare_topk(g["ids_k100"], g["scores_k100"], res.ids.cpu().numpy(), res.scores.cpu().numpy(), rtol=RTOL)
    v = idx.vector                                                        # exported back from the device format
    assert v.layout == torch.sparse_csr and (v.col_indices().cpu().numpy() == ix).all() and (v.values().cpu().numpy() == d).all()
    assert "SparseIndex" in str(idx) and "torch.sparse_csr" in str(idx) and "2000, 29523" in str(idx)
    with pytest.raises(RuntimeError):
        idx.search(q, 2001)


def test_dense_index_search_golden(golden):
    g = golden("search_dense")
    ip, ix, d = oracle.synth_csr(0, 0, 2000)
    idx = Index()
    idx.vector = csr_tensor(ip, ix, d).to_dense()
    idx.move_to_device("cuda")
    res = idx.search(torch.from_numpy(oracle.synth_queries(1, 8)), 100)
    compare.compare_topk(g["ids_k100"], g["scores_k100"], res.ids.cpu().numpy(), res.scores.cpu().numpy(), rtol=RTOL)


def test_retrieve_and_rerank_golden(golden):
    g = golden("retrieve")                                                # retriever.py:107-148 run unbound on a fake self
    n, b, k = int(g["n"]), int(g["b"]), int(g["k"])
    s_bot, s_par, s_q = g["seeds"].tolist()
    ip, ix, d = oracle.synth_csr(s_bot, 0, n, V, 86, synth.KIND_BOT)
    bot = BoTIndex()
    bot.data = [str(i) for i in range(n)]
    bot.vector = csr_tensor(ip, ix, d.astype(np.float32))
    bot.move_to_device("cuda")
    ip2, ix2, d2 = oracle.synth_csr(s_par, 0, n)
    p_dense = csr_tensor(ip2, ix2, d2).to_dense().cuda()

    def fake_embed(texts, batch_size=32, require_grad=False, **kw):
        return p_dense[[int(t) for t in texts]]

    fake = types.SimpleNamespace(index=bot, device="cuda", encoder_q=types.SimpleNamespace(config=types.SimpleNamespace(topk=768)),
                                 encoder_p=types.SimpleNamespace(embed=fake_embed))
    fake.process_query = types.MethodType(Retriever.process_query, fake)
    fake._rerank = types.MethodType(Retriever._rerank, fake)
    q = oracle.synth_queries(s_q, b)
    r_t = Retriever.retrieve(fake, torch.from_numpy(q), k=k)
    r_n = Retriever.retrieve(fake, q, k=k)                                # ndarray queries (retriever.py:96-97)
    assert (r_t.ids == r_n.ids).all()
    compare.compare_topk(g["ids"], g["scores"], r_t.ids.cpu().numpy(), r_t.scores.cpu().numpy(), rtol=RTOL)
    r_r = Retriever.retrieve(fake, torch.from_numpy(q), k=k, rerank=True)
    assert r_r.ids.is_cuda and r_r.scores.is_cuda                         # scored and sorted on the device (vs_rerank_scores / vs_rerank_topk)
    compare.compare_topk(g["rerank_ids"], g["rerank_scores"], r_r.ids.cpu().numpy(), r_r.scores.cpu().numpy(), rtol=RTOL)
    # the same through a small re-embedding batch (several vs_rerank_scores calls per query batch)
    r_r2 = Retriever.retrieve(fake, torch.from_numpy(q), k=k, rerank=True, batch_size=1)
    assert (r_r2.ids == r_r.ids).all() and (r_r2.scores == r_r.scores).all()
    sparse = SparseIndex()
    sparse.data = bot.data
    sparse.vector = csr_tensor(ip2, ix2, d2)
    sparse.move_to_device("cuda")
    r_s = Retriever.retrieve(fake, torch.from_numpy(q), k=k, rerank=True, index=sparse)   # passed index is honoured; rerank ignored
    compare.compare_topk(g["sparse_ids"], g["sparse_scores"], r_s.ids.cpu().numpy(), r_s.scores.cpu().numpy(), rtol=RTOL)
    with pytest.raises(NotImplementedError):
        Retriever.process_query(fake, 3.14)


def test_bot_index_fp16_dtype_and_exact_ids():
    n = 3000
    ip, ix, _ = oracle.synth_csr(3, 0, n, V, 86, synth.KIND_BOT)
    idx = BoTIndex()
    idx.vector = torch.sparse_csr_tensor(torch.from_numpy(ip), torch.from_numpy(ix.astype(np.int64)),
                                         torch.ones(len(ix), dtype=torch.float16), size=(n, V))   # what _build_bot_vectors returns
    idx.move_to_device("cuda")
    q = oracle.synth_queries(5, 6, val_law=synth.VAL_DYADIC)
    res = idx.search(torch.from_numpy(q), 100)
    assert res.scores.dtype == torch.float16                             # scores come back in the index dtype (index.py:89-93)
    o_ids, o_sc = oracle.csr_search(ip, ix, None, V, q.astype(np.float16).astype(np.float32), 100)
    assert (res.ids.cpu().numpy() == o_ids).all()
    assert (res.scores.cpu().numpy() == o_sc.astype(np.float16)).all()
    # a BoTIndex over a VALUED matrix: the reference searches it like any sparse index (index.py:205-218) -- so does the drop-in
    ipv, ixv, dv = oracle.synth_csr(4, 0, 500)
    val = BoTIndex()
    val.vector = torch.sparse_csr_tensor(torch.from_numpy(ipv), torch.from_numpy(ixv.astype(np.int64)), torch.from_numpy(dv), size=(500, V))
    val.move_to_device("cuda")
    qv = oracle.synth_queries(6, 5)
    res = val.search(torch.from_numpy(qv), 50)
    o_ids, o_sc = oracle.csr_search(ipv, ixv, dv, V, qv, 50, acc64=True)
    compare.compare_topk(o_ids, o_sc, res.ids.cpu().numpy(), res.scores.float().cpu().numpy(), rtol=1e-4)


def test_save_load_roundtrip(golden, tmp_path):
    g = golden("save_load")                                               # index.py:163-202
    n = 10
    ip, ix, d = oracle.synth_csr(int(g["seed"]), 0, 2 * n)
    for s in range(2):
        sl = slice(ip[s * n], ip[(s + 1) * n])
        idx = SparseIndex(device="cuda")
        idx.vector = csr_tensor(ip[s * n:(s + 1) * n + 1] - ip[s * n], ix[sl], d[sl])
        idx.move_to_device("cuda")
        idx.save(str(tmp_path / f"index{s}.npz"))
    with np.load(tmp_path / "index0.npz") as z:
        assert sorted(z.files) == g["manifest_keys"].tolist()
        assert [str(z[k].dtype) for k in sorted(z.files)] == g["manifest_dtypes"].tolist()
        assert (z["indices"] == g["reload_indices"]).all() and (z["data"] == g["reload_data"]).all() and (z["indptr"] == g["reload_indptr"]).all()
    for tag, shift in (("shift0", 0), ("shift999", 999)):
        li = SparseIndex(str(tmp_path / "index*.npz"), None, fp16=False, device="cuda", shift=shift)
        v = li.vector
        assert list(v.shape) == g[f"{tag}_shape"].tolist()
        assert (v.crow_indices().cpu().numpy() == g[f"{tag}_indptr"]).all()
        assert (v.col_indices().cpu().numpy() == g[f"{tag}_indices"]).all()
        assert (v.values().cpu().numpy() == g[f"{tag}_data"]).all()
    # fp16=True (the upstream default) is applied on the device
    l16 = SparseIndex(str(tmp_path / "index*.npz"), None, device="cuda")
    assert l16.vector.values().dtype == torch.float16
    assert (l16.vector.values().cpu().numpy() == d.astype(np.float16)).all()
    q = torch.from_numpy(oracle.synth_queries(2, 3))
    r = l16.search(q, 5)
    assert r.scores.dtype == torch.float16
    o_ids, o_sc = oracle.csr_search(ip, ix, d.astype(np.float16).astype(np.float32), V, q.numpy().astype(np.float16).astype(np.float32), 5, acc64=True)
    compare.compare_topk(o_ids, o_sc.astype(np.float16).astype(np.float32), r.ids.cpu().numpy(), r.scores.float().cpu().numpy(), rtol=1e-3)
    # Retriever.load_index infers the type from the extension and accepts IndexType (upstream: str only)
    rt = Retriever.__new__(Retriever)
    torch.nn.Module.__init__(rt)
    rt._dummy = torch.nn.Parameter(torch.zeros(1, device="cuda"))
    Retriever.load_index(rt, index_file=str(tmp_path / "index0.npz"))
    assert isinstance(rt.index, SparseIndex) and rt.index_type == IndexType.SPARSE
    Retriever.load_index(rt, index_file=str(tmp_path / "index0.npz"), index_type=IndexType.SPARSE)
    assert isinstance(rt.index, SparseIndex)
    # dense .pt shards (broken upstream, index.py:36-44)
    dense = csr_tensor(ip, ix, d).to_dense()
    torch.save(dense[:n].clone(), tmp_path / "d0.pt")
    torch.save(dense[n:].clone(), tmp_path / "d1.pt")
    di = Index(str(tmp_path / "d*.pt"), None, fp16=False, device="cuda")
    assert (di.vector.cpu() == dense).all()
    di.save(str(tmp_path / "all.pt"))
    assert (torch.load(tmp_path / "all.pt") == dense).all()


# ---- encoder + build_index on a small random-init BERT -------------------------------------------------
class FakeTokenizer:
    """Whitespace 'tokenizer': texts are space-separated token ids (no WordPiece vocab offline)."""
    vocab = range(VOCAB)

    def _ids(self, texts, max_length, truncation):
        out = []
        for t in texts:
            ids = [int(x) for x in t.split()]
            if truncation and max_length and len(ids) > max_length:
                ids = ids[:max_length - 1] + [102]
            out.append(ids)
        return out

    def __call__(self, texts, max_length=None, truncation=False):
        return {"input_ids": self._ids(texts, max_length, truncation)}

    def batch_encode_plus(self, texts, padding=True, truncation=True, max_length=None, return_tensors="pt"):
        from transformers import BatchEncoding
        rows = self._ids(texts, max_length, truncation)
        L = max(len(r) for r in rows)
        ids = torch.tensor([r + [0] * (L - len(r)) for r in rows])
        mask = torch.tensor([[1] * len(r) + [0] * (L - len(r)) for r in rows])
        return BatchEncoding({"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask})

    def convert_ids_to_tokens(self, ids):
        return [f"tok{i}" for i in ids]


def make_texts(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        body = rng.integers(1996, 9000, size=int(rng.integers(5, 40)))
        out.append(" ".join(map(str, [101] + body.tolist() + [102])))
    return out


@pytest.fixture(scope="module")
def tiny_retriever():
    from vsearch_amd.ir import RetrieverConfig
    from vsearch_amd.ir.encoder.vdr import VDREncoder, VDREncoderConfig
    torch.manual_seed(0)
    kw = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, vocab_size=VOCAB,
              max_len=48, topk=32, random_init=True, type="vdr")
    enc_q = VDREncoder(VDREncoderConfig(**kw), tokenizer=FakeTokenizer())
    enc_p = VDREncoder(VDREncoderConfig(**kw), tokenizer=FakeTokenizer())
    r = Retriever(RetrieverConfig(encoder_q=kw, encoder_p=kw), encoder_q=enc_q, encoder_p=enc_p)
    return r.to("cuda").eval()


def reference_embed(enc, texts, topk, activate_lexical=True, bow=False, max_len=None):
    """vdr.py:97-179 restated with plain torch ops on the same weights (the torch fp32 reference for the HIP head)."""
    import torch.nn.functional as F
    encd = enc.encode(texts, max_len=max_len)
    ids = encd["input_ids"]
    bow_mask = torch.zeros([ids.shape[0], VOCAB], device=ids.device).scatter_(-1, ids, 1).bool().float()[:, SHIFT:]
    if bow:
        return bow_mask
    with torch.no_grad():
        h = enc.ln(enc.bert_model(**encd).last_hidden_state)
        emb = (F.elu(h @ enc.bert_model.embeddings.word_embeddings.weight[SHIFT:].t()) + 1).max(1)[0]
    if topk == 0:
        tk = torch.zeros_like(emb)
    elif topk in (None, -1):
        tk = torch.ones_like(emb)
    else:
        tk = torch.zeros_like(emb, dtype=torch.bool).scatter_(-1, emb.topk(topk).indices, True)
    mask = torch.logical_or(bow_mask, tk) if activate_lexical else tk
    return emb * mask


@pytest.mark.parametrize("kw", [dict(topk=32), dict(topk=32, activate_lexical=False), dict(topk=0), dict(topk=-1, activate_lexical=False), dict(bow=True)])
def test_encoder_embed_matches_torch_reference(tiny_retriever, kw):
    enc = tiny_retriever.encoder_q
    texts = make_texts(7, 1)
    got = enc.embed(texts, batch_size=3, **kw)
    want = torch.cat([reference_embed(enc, texts[s:s + 3], kw.get("topk", 32), kw.get("activate_lexical", True), kw.get("bow", False))
                      for s in range(0, 7, 3)])
    assert got.shape == (7, V) and got.is_cuda
    assert ((got != 0) == (want != 0)).all()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    assert isinstance(enc.embed(texts[0], convert_to_tensor=False, **kw), np.ndarray)
    with pytest.raises(NotImplementedError):
        enc.embed(texts, require_grad=True)


@pytest.mark.parametrize("index_type", ["sparse", IndexType.DENSE, IndexType.BAG_OF_TOKEN])
def test_build_index_and_retrieve_text_queries(tiny_retriever, index_type, tmp_path):
    r = tiny_retriever
    texts = make_texts(50, 2)
    r.build_index(texts, batch_size=16, index_type=index_type)
    assert len(r.index) == 50 and r.index.get_sample(3) == texts[3]
    p_ref = torch.cat([reference_embed(r.encoder_p, texts[s:s + 16], 32, activate_lexical=False, max_len=128) for s in range(0, 50, 16)])
    v = r.index.vector
    if r.index.index_type == IndexType.BAG_OF_TOKEN:
        toks = [[int(x) for x in t.split()] for t in texts]
        o_ip, o_ix = oracle.bot_build(toks, VOCAB, SHIFT)
        assert v.values().dtype == torch.float16 and (v.crow_indices().cpu().numpy() == o_ip).all() and (v.col_indices().cpu().numpy() == o_ix).all()
        p_ref = (v.to_dense().float() != 0).float().cuda()
    elif r.index.index_type == IndexType.SPARSE:
        assert v.layout == torch.sparse_csr and ((v.to_dense() != 0).cuda() == (p_ref != 0)).all()
        assert (torch.diff(v.crow_indices()) == 32).all()                  # exactly topk nnz per passage (activate_lexical=False)
    else:
        torch.testing.assert_close(v.cuda(), p_ref, rtol=1e-5, atol=1e-6)
    queries = make_texts(5, 3)
    res = r.retrieve(queries, k=10, a=32)
    q_ref = reference_embed(r.encoder_q, queries, 32, activate_lexical=True)
    want = (q_ref.double() @ p_ref.double().t()).float().cpu().numpy()
    ids, sc = res.ids.cpu().numpy(), res.scores.float().cpu().numpy()
    compare.check_topk_valid(want, ids, sc, rtol=1e-3 if r.index.index_type == IndexType.BAG_OF_TOKEN else RTOL)
    one = r.retrieve(queries[0], k=3)                                     # single str query
    assert tuple(one.ids.shape) == (1, 3)
    path = str(tmp_path / ("idx.pt" if r.index.index_type == IndexType.DENSE else "idx.npz"))
    r.save_index(path)
    assert os.path.getsize(path) > 0


@pytest.mark.parametrize("B,L,H,vocab", [(4, 16, 768, 3000), (3, 200, 64, 2500), (2, 129, 96, 1200), (5, 7, 32, 999 + 130)])
def test_fused_head_project_pool(golden, B, L, H, vocab):
    """vs_head_project_pool == elu1p(max_l LN(h) @ W[shift:].T) (vdr.py:71-75), incl. the golden encoder-head fixture."""
    shift = 999
    if (B, L, H, vocab) == (4, 16, 768, 3000):
        g = golden("encoder_head")
        s = g["seeds"].tolist()
        hidden = torch.from_numpy(synth.dense_uniform(s[0], (B, L, H), -2.0, 2.0)).cuda()
        W = torch.from_numpy(synth.dense_uniform(s[1], (vocab, H), -0.08, 0.08)).cuda()
        ln = torch.nn.LayerNorm(H).cuda()
        with torch.no_grad():
            ln.weight.copy_(torch.from_numpy(synth.dense_uniform(s[2], (H,), 0.5, 1.5)))
            ln.bias.copy_(torch.from_numpy(synth.dense_uniform(s[3], (H,), -0.1, 0.1)))
            h_ln = ln(hidden)
        want_golden = g["emb"]
    else:
        gen = torch.Generator(device="cuda").manual_seed(B * 1000 + L)
        h_ln = torch.randn((B, L, H), device="cuda", generator=gen)
        W = torch.randn((vocab, H), device="cuda", generator=gen) * 0.1
        want_golden = None
    got = sp.head_project_pool(h_ln, W[shift:])
    want = torch.nn.functional.elu(h_ln.double() @ W[shift:].double().t()).add(1).max(1)[0].float()
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-6)
    if want_golden is not None:
        np.testing.assert_allclose(got.cpu().numpy(), want_golden, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("Vv", [29523, 8000, 33000], ids=["vdr-vocabulary", "two-load-rows", "beyond-the-register-kernel"])
def test_embed_mask_signed_and_special_values(Vv):
    """The mask stage on rows of continuous values with negatives, both zeros and infinities (vdr.py:152-169: emb *= mask, so an
    unselected -3.5 becomes -0.0 and an unselected inf a NaN): mask_rows_fast_kernel (V <= 32 Ki: key halves in registers / LDS,
    elements rebuilt from them) and mask_rows_kernel (beyond) against the oracle, bit patterns included; 4 rows per workgroup and more
    rows than workgroups exercise the prefetch of the next row."""
    rng = np.random.default_rng(Vv)
    B, L, shift = 1030, 24, 999
    vocab = Vv + shift
    emb = (rng.standard_normal((B, Vv)) * 3).astype(np.float32)
    emb[:, ::7] = np.abs(emb[:, ::7])
    emb[3, :50] = 0.0
    emb[3, 50:90] = -0.0
    emb[5, 10] = np.inf
    emb[5, 11] = -np.inf
    emb[6, :] = -np.abs(emb[6, :])                                       # a row of negatives only
    emb[7, :] = 0.25                                                     # one value: every tie goes to the lowest columns
    ids = rng.integers(0, vocab, size=(B, L)).astype(np.int64)
    for topk, lex in ((768, True), (1, False), (Vv, False), (Vv // 2, True)):
        want = oracle.embed_mask(emb, ids, vocab, shift, topk=topk, activate_lexical=lex)
        got = torch.from_numpy(emb).cuda()
        sp.apply_embed_mask_(got, torch.from_numpy(ids), vocab, shift, topk, lex)
        got = got.cpu().numpy()
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (Vv, topk, lex, np.argwhere(~same)[:5])
    k = 777
    assert (sp.build_topk_mask(torch.from_numpy(emb), k).numpy() == oracle.topk_mask(emb, k)).all()


@pytest.mark.parametrize("seed", range(5))
def test_random_embed_mask_against_oracle(seed):
    """Randomised mask stage (vdr.py:152-169): heavy ties (few distinct values), odd vocab sizes, every flag combination."""
    rng = np.random.default_rng(seed)
    vocab = int(rng.choice([1200, 4000, 30522]))
    shift = int(rng.choice([0, 999]))
    Vv = vocab - shift
    B, L = int(rng.integers(1, 9)), int(rng.integers(1, 40))
    ids = rng.integers(0, vocab, size=(B, L)).astype(np.int64)
    emb = rng.integers(0, 12, size=(B, Vv)).astype(np.float32) / 4           # many exact ties at the k-th value
    for topk in (0, 1, 17, Vv // 3, -1):
        for lex in (False, True):
            want = oracle.embed_mask(emb, ids, vocab, shift, topk=topk, activate_lexical=lex)
            got = torch.from_numpy(emb).cuda()
            sp.apply_embed_mask_(got, torch.from_numpy(ids), vocab, shift, topk, lex)
            assert (got.cpu().numpy() == want).all(), (seed, topk, lex)
    want = oracle.embed_mask(emb, ids, vocab, shift, bow=True)
    got = torch.from_numpy(emb).cuda()
    sp.apply_embed_mask_(got, torch.from_numpy(ids), vocab, shift, 0, True, bow=True)
    assert (got.cpu().numpy() == want).all()
    for norm in (False, True):
        np.testing.assert_allclose(sp.build_bow_mask(torch.from_numpy(ids), vocab, shift, norm).numpy(), oracle.bow_mask(ids, vocab, shift, norm), rtol=1e-6)
    k = int(rng.integers(0, Vv + 1))
    assert (sp.build_topk_mask(torch.from_numpy(emb), k).numpy() == oracle.topk_mask(emb, k)).all()
    rp, ci, va = sp.dense_to_csr(torch.from_numpy(emb).cuda())
    ref = torch.from_numpy(emb).to_sparse_csr()
    assert (rp.cpu() == ref.crow_indices()).all() and (ci.cpu() == ref.col_indices()).all() and (va.cpu() == ref.values()).all()


def test_native_shard_file_roundtrip(tmp_path):
    """.vsx = the device format verbatim: save -> load gives the same index (values, search results) without scipy."""
    n = 700
    ip, ix, d = oracle.synth_csr(17, 0, n, V, 300)
    idx = SparseIndex(device="cuda")
    idx.vector = csr_tensor(ip, ix, d)
    idx.move_to_device("cuda")
    idx.save(str(tmp_path / "a.vsx"))
    back = SparseIndex(str(tmp_path / "a.vsx"), None, device="cuda")
    v = back.vector
    assert (v.crow_indices().cpu().numpy() == ip).all() and (v.col_indices().cpu().numpy() == ix).all() and (v.values().cpu().numpy() == d).all()
    q = torch.from_numpy(oracle.synth_queries(1, 4))
    a, b = idx.search(q, 20), back.search(q, 20)
    assert (a.ids == b.ids).all() and (a.scores == b.scores).all()
    bot = BoTIndex(device="cuda")
    ipb, ixb, _ = oracle.synth_csr(3, 0, 500, V, 86, synth.KIND_BOT)
    bot.vector = torch.sparse_csr_tensor(torch.from_numpy(ipb), torch.from_numpy(ixb.astype(np.int64)), torch.ones(len(ixb), dtype=torch.float16), size=(500, V))
    bot.move_to_device("cuda")
    bot.save(str(tmp_path / "b.vsx"))
    bb = BoTIndex(str(tmp_path / "b.vsx"), None, device="cuda")
    assert (bb.search(q, 10).ids == bot.search(q, 10).ids).all() and bb.vector.values().dtype == torch.float16
    (tmp_path / "bad.vsx").write_bytes(b"not a shard file")
    with pytest.raises(ValueError, match="native shard"):
        SparseIndex(str(tmp_path / "bad.vsx"), None, device="cuda")
    # the file is searched as it is: the loader checks what the scan kernels rely on
    raw = (tmp_path / "a.vsx").read_bytes()
    hdr = 8 + 4 + 4 + 8 + 8 + 8 + 32                                     # magic, store_dtype, n_cols, n_rows, n_packets, nnz, reserved
    n_pk = int(np.frombuffer(raw, dtype=np.int64, count=1, offset=24)[0])
    cases = {
        "truncated": raw[:-100],                                                                   # payload shorter than the header says
        "padded": raw + b"\0" * 64,
        "rowptr": raw[:hdr + 8] + (2 ** 31).to_bytes(4, "little") + raw[hdr + 12:],                # pk_ptr[2] beyond the packets
        "column": raw[:hdr + (n + 1) * 4] + (V + 5).to_bytes(2, "little") + raw[hdr + (n + 1) * 4 + 2:],   # a column id above n_cols
        "pad-inside": raw[:hdr + (n + 1) * 4] + V.to_bytes(2, "little") + raw[hdr + (n + 1) * 4 + 2:],     # a pad column at a row's start
        "nnz": raw[:32] + (int(ip[-1]) + 8).to_bytes(8, "little") + raw[40:],                       # header nnz != payload
    }
    assert n_pk * 16 < len(raw)
    for name, blob in cases.items():
        (tmp_path / f"{name}.vsx").write_bytes(blob)
        with pytest.raises(ValueError):
            SparseIndex(str(tmp_path / f"{name}.vsx"), None, device="cuda")
            pytest.fail(f"corrupt file '{name}' was accepted")
    # no binary file into a SparseIndex, no shift; a BoTIndex takes a valued file (the reference's BoTIndex searches any sparse matrix)
    assert (BoTIndex(str(tmp_path / "a.vsx"), None, device="cuda").search(q, 20).ids == a.ids).all()
    with pytest.raises(ValueError, match="binary"):
        SparseIndex(str(tmp_path / "b.vsx"), None, device="cuda")
    with pytest.raises(ValueError, match="shift"):
        SparseIndex(str(tmp_path / "a.vsx"), None, device="cuda", shift=999)


@pytest.mark.parametrize("topk,lexical", [(768, False), (768, True), (32, True), (1, False), (5000, False)])
def test_embed_mask_to_csr_equals_mask_then_to_sparse_csr(topk, lexical):
    """SURVEY 8(f1) / VERDICT r4 item 7: the mask stage fused with to_sparse_csr() (vs_embed_mask_to_csr) returns exactly the CSR of the
    masked dense batch (vs_embed_mask + vs_dense_to_csr; vdr.py:152-169 + retriever.py:304) -- row pointers, columns and value bits --
    on rows with ties, exact zeros among the selected, negative values and a ragged batch; the input is not modified."""
    g = torch.Generator().manual_seed(11)
    B, L = 37, 23
    emb = torch.rand((B, V), generator=g) * 3
    emb[3] = 0.0                                                      # an all-zero row: top-k selects zeros, the CSR row is empty
    emb[4, ::7] = 1.5                                                 # ties across the k-th value
    emb[5] = -emb[5]                                                  # negative values (selected by rank, kept as they are)
    emb[6, :100] = 0.0
    emb = emb.cuda()
    ids = torch.randint(SHIFT, VOCAB, (B, L), generator=g).cuda()
    ids[7, :5] = torch.tensor([0, 101, 102, 999, VOCAB - 1])          # tokens below the shift are dropped
    keep = emb.clone()
    rp, ci, va = sp.embed_mask_to_csr(emb, ids, VOCAB, SHIFT, topk, lexical)
    assert (emb == keep).all()
    ref = emb.clone()
    sp.apply_embed_mask_(ref, ids if lexical else None, VOCAB, SHIFT, topk, lexical)
    r_rp, r_ci, r_va = sp.dense_to_csr(ref)
    assert (rp == r_rp).all() and (ci == r_ci).all() and (va.view(torch.int32) == r_va.view(torch.int32)).all()
    with pytest.raises(NotImplementedError):
        sp.embed_mask_to_csr(emb, ids, VOCAB, SHIFT, 0, True)
    with pytest.raises(NotImplementedError):                          # more kept elements a row than the fused kernel stages: the two-call path serves those
        sp.embed_mask_to_csr(emb, ids, VOCAB, SHIFT, 8192, True)


@pytest.mark.parametrize("B,topk,lexical", [(600, 768, True), (257, 64, False), (1500, 8000, True)])
def test_embed_mask_to_csr_many_rows_per_workgroup(B, topk, lexical):
    """The fused kernel gives every workgroup a run of consecutive rows (uneven when B is not a multiple of the CU count), ranks a row's
    kept elements with wave scans and moves the workgroup's run to its place once the workgroups before it have published their totals:
    row pointers, columns and value bits equal mask-then-to_sparse_csr on batches of 1 .. 6 rows a workgroup, with zero rows, ties and
    lexical columns whose value is zero in the middle of a run."""
    g = torch.Generator().manual_seed(B)
    L = 40
    emb = torch.rand((B, V), generator=g) * 3
    emb[B // 2] = 0.0
    emb[B // 3, ::5] = 2.0
    emb[B - 1, 1000:] = 0.0
    emb[7, ::2] = 0.0
    emb = emb.cuda()
    ids = torch.randint(SHIFT, VOCAB, (B, L), generator=g).cuda()
    rp, ci, va = sp.embed_mask_to_csr(emb, ids, VOCAB, SHIFT, topk, lexical)
    ref = emb.clone()
    sp.apply_embed_mask_(ref, ids if lexical else None, VOCAB, SHIFT, topk, lexical)
    r_rp, r_ci, r_va = sp.dense_to_csr(ref)
    assert (rp == r_rp).all() and (ci == r_ci).all() and (va.view(torch.int32) == r_va.view(torch.int32)).all()


def test_encoder_embed_csr_equals_embed_then_to_sparse_csr(tiny_retriever):
    enc = tiny_retriever.encoder_p
    texts = make_texts(19, 3)
    dense = enc.embed(texts, batch_size=8, activate_lexical=False)
    rows = 0
    for (rp, ci, va, n_cols), s in zip(enc.embed_csr(texts, batch_size=8, activate_lexical=False), range(0, 19, 8)):
        r_rp, r_ci, r_va = sp.dense_to_csr(dense[s:s + 8].contiguous())
        assert n_cols == V and (rp == r_rp).all() and (ci == r_ci).all() and (va == r_va).all()
        rows += rp.numel() - 1
    assert rows == 19


def test_mean_pooling_with_pooling_topk():
    """vdr.py:76-79 (pooling = "mean" with pooling_topk): mean of the t largest elu1p activations per vocabulary dimension."""
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn((3, 37, 1000), generator=g) * 2).cuda()
    for t in (1, 4, 32):
        got = sp.head_pool_mean_topk(logits, t)
        act = torch.where(logits > 0, logits + 1, torch.exp(logits))          # elu1p (sparse.py:6)
        want = act.topk(t, dim=1).values.mean(1)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        sp.head_pool_mean_topk(logits[:, :8], 9)


# ---- row shards over several GPUs through the reference's API --------------------------------------------
def _write_shard_files(tmp_path, n_files, rows_per_file, binary=False, seed=11):
    """`n_files` scipy .npz shards of one synthetic corpus (the reference's per-shard build, examples/inference_sparse/README.md:90-107)"""
    from scipy.sparse import csr_array, save_npz
    n = n_files * rows_per_file
    ip, ix, d = oracle.synth_csr(seed, 0, n, V, 86 if binary else 768, 1 if binary else 0)
    if binary:
        d = np.ones(len(ix), dtype=np.float32)
    for s in range(n_files):
        r0, r1 = s * rows_per_file, (s + 1) * rows_per_file
        sl = slice(ip[r0], ip[r1])
        save_npz(tmp_path / f"shard{s:02d}.npz", csr_array((d[sl], ix[sl].astype(np.int64), (ip[r0:r1 + 1] - ip[r0]).astype(np.int64)), shape=(rows_per_file, V)))
    return ip, ix, d


@pytest.mark.parametrize("n_files,n_devices", [(2, 2), (8, 8), (8, 3)])
def test_sparse_index_row_sharded_over_devices_equals_the_unsharded_index(tmp_path, n_files, n_devices):
    """SparseIndex(index_file="shard*.npz", devices=[...]): the shard files are dealt to the GPUs in row order (here: all the same
    GPU -- the box has one), search goes through vs_shard_group_*: ids and scores equal the unsharded facade's bit for bit."""
    rows = 24_000 // n_files
    ip, ix, d = _write_shard_files(tmp_path, n_files, rows)
    pattern = str(tmp_path / "shard*.npz")
    one = SparseIndex(pattern, None, fp16=False, device="cuda")
    many = SparseIndex(pattern, None, fp16=False, device="cuda", devices=[0] * n_devices)
    assert many.shards is not None and len(many.shards) == min(n_files, n_devices) and one.shards is None
    assert sum(s.info().n_rows for s in many.shards) == n_files * rows
    q = torch.from_numpy(oracle.synth_queries(3, 16))
    a, b = one.search(q, 100), many.search(q, 100)
    assert (a.ids.cpu() == b.ids.cpu()).all() and (a.scores.cpu() == b.scores.cpu()).all()
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q.numpy(), 100, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, b.ids.cpu().numpy(), b.scores.float().cpu().numpy(), rtol=RTOL)
    # the `vector` of a sharded index is the shards' rows re-joined (index.py:175)
    v = many.vector
    assert tuple(v.shape) == (n_files * rows, V) and (v.crow_indices().cpu().numpy() == ip).all()
    # Retriever.load_index(devices=...) builds the same thing
    rt = Retriever.__new__(Retriever)
    torch.nn.Module.__init__(rt)
    rt._dummy = torch.nn.Parameter(torch.zeros(1, device="cuda"))
    Retriever.load_index(rt, index_file=pattern, devices=[0, 0])
    assert isinstance(rt.index, SparseIndex) and len(rt.index.shards) == 2
    c = rt.index.search(q.to(torch.float16), 100)                       # (load_index keeps the reference's fp16=True default)
    assert c.scores.dtype == torch.float16


@pytest.mark.parametrize("kind,n_files,n_devices", [("npz", 1, 3), ("npz", 2, 3), ("vsx", 1, 2), ("vsx", 2, 3)])
def test_row_range_sharding_of_fewer_files_than_devices(tmp_path, kind, n_files, n_devices):
    """VERDICT r4 item 6a ("per-shard npz or row ranges"): ONE .npz -- what SparseIndex.save writes, index.py:181-202 --
    or fewer files than GPUs: the rows are dealt in contiguous ranges (vs_index_slice_rows, device-to-device); bit-equal to unsharded."""
    rows = 9_000 // n_files
    ip, ix, d = _write_shard_files(tmp_path, n_files, rows)
    pattern = str(tmp_path / "shard*.npz")
    one = SparseIndex(pattern, None, fp16=False, device="cuda")
    if kind == "vsx":
        for i in range(n_files):
            part = SparseIndex(str(tmp_path / f"shard{i:02d}.npz"), None, fp16=False, device="cuda")
            part.save(str(tmp_path / f"part{i:02d}.vsx"))
        pattern = str(tmp_path / "part*.vsx")
    many = SparseIndex(pattern, None, fp16=False, device="cuda", devices=[0] * n_devices)
    assert many.shards is not None and len(many.shards) >= n_devices
    assert sum(s.info().n_rows for s in many.shards) == n_files * rows
    assert sum(s.info().nnz for s in many.shards) == len(ix)
    q = torch.from_numpy(oracle.synth_queries(3, 16))
    a, b = one.search(q, 100), many.search(q, 100)
    assert (a.ids.cpu() == b.ids.cpu()).all() and (a.scores.cpu() == b.scores.cpu()).all()
    v = many.vector
    assert tuple(v.shape) == (n_files * rows, V) and (v.crow_indices().cpu().numpy() == ip).all() and (v.col_indices().cpu().numpy() == ix).all()


def test_build_index_with_devices_deals_row_ranges(tiny_retriever):
    """Retriever.build_index(..., devices=) (retriever.py:284-317 builds ONE in-memory matrix): row ranges over the GPUs, same results."""
    texts = make_texts(41, 5)
    tiny_retriever.build_index(texts, batch_size=8, index_type="sparse")
    one = tiny_retriever.index
    q = tiny_retriever.encoder_q.embed(make_texts(5, 6))
    a = one.search(q, 10)
    tiny_retriever.build_index(texts, batch_size=8, index_type="sparse", devices=[0, 0, 0])
    many = tiny_retriever.index
    assert many.shards is not None and len(many.shards) == 3 and sum(s.info().n_rows for s in many.shards) == 41
    b = many.search(q, 10)
    assert (a.ids.cpu() == b.ids.cpu()).all() and (a.scores.cpu() == b.scores.cpu()).all()
    res = tiny_retriever.retrieve(make_texts(2, 7), k=3)
    assert len(res.ids) == 2


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: lights up by itself on the first multi-GPU box (VERDICT r4 item 6b)")
def test_shard_group_and_rccl_bench_across_real_devices(tmp_path):
    """On a box with >= 2 GPUs: (a) vs_shard_group_search with one shard per physical GPU (peer copies over xGMI) and (b) `bench.py
    --gpus 2` (one process per GPU, RCCL all-gather) against the unsharded result, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    from vsearch_amd import _native as nat
    from vsearch_amd.device_index import DeviceIndex, ShardGroup
    from vsearch_amd.distributed import shard_rows
    g = min(torch.cuda.device_count(), 8)
    n = 40_000 * g
    whole = DeviceIndex.synthetic(0, 0, n, V, 768, 0, 0, nat.VS_F32, 0)
    parts = [DeviceIndex.synthetic(0, *shard_rows(n, g, r), V, 768, 0, 0, nat.VS_F32, r) for r in range(g)]
    sliced = [whole.slice_rows(*shard_rows(n, g, r), r) for r in range(g)]          # peer copies of the packets
    q = torch.from_numpy(oracle.synth_queries(1, 32)).cuda(0)
    ids, sc = whole.search(q, 100)
    for shards in (parts, sliced):
        grp = ShardGroup(shards)
        g_ids, g_sc = grp.search(q, 100)
        assert (g_ids.cpu() == ids.cpu()).all() and (g_sc.cpu() == sc.cpu()).all()
        grp.close()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = {}
    for gpus in (1, 2):
        dump = str(tmp_path / f"ids{gpus}.npz")
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(gpus), "--docs", "400000", "--steps", "1", "--warmup", "0", "--batch", "64",
                            "--no-cpu-baseline", "--no-secondary", "--dump-ids", dump], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == gpus
        if gpus == 2:
            assert line["exchange"]["process_group_backend"] == "nccl"
        outs[gpus] = np.load(dump)
    assert (outs[1]["ids"] == outs[2]["ids"]).all() and (outs[1]["scores"] == outs[2]["scores"]).all()


def test_lock_step_wait_times_out_instead_of_hanging():
    """the lock step of the walks (pace_wait, bp_walk.h) is a bounded wait.  VS_BP_KNOB=64 makes workgroup 0 of every walk
    launch never report its progress: every peer's wait must time out (~4 ms, once) and the launch must finish with the right result
    -- on the quad walk (valued index) and the bag-of-token walk.  A subprocess: the library reads the knob once."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import numpy as np, torch, sys
sys.path.insert(0, %r)
import oracle
from vsearch_amd import _native as nat, synth
from vsearch_amd.device_index import DeviceIndex
for kind, nnz, store, law in ((0, 768, nat.VS_F32, 0), (synth.KIND_BOT, 86, nat.VS_NONE, synth.VAL_DYADIC)):
    idx = DeviceIndex.synthetic(5, 0, 150_000, 29523, nnz, kind, 0, store)
    q = torch.from_numpy(oracle.synth_queries(1, 256, 29523, 776, law)).cuda()
    idx.set_option("blocked_postings", 0)
    ids0, sc0 = idx.search(q, 100)
    idx.set_option("blocked_postings", 1)
    idx.set_option("postings_pace", 4)
    ids1, sc1 = idx.search(q, 100)
    assert idx.info().last_path == 3
    assert (ids0.cpu() == ids1.cpu()).all() and (sc0.cpu() == sc1.cpu()).all()
print("OK")
""" % repo
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VS_BP_KNOB="64"))
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_a_late_wave_does_not_split_the_cut_decision():
    """Round 6 (docs/EXPERIMENTS.md, profiles/r06_prune_decision_race.txt): whether a candidate buffer is cut after an epilogue round is ONE
    decision of the workgroup -- taken by the wave that arrives last at the round's end.  Until round 6 every thread read the counters behind
    the barrier; a wave that read them late saw the next round's pushes and went into the cut's barriers alone (whole blocks of candidates
    lost once other processes' waves on the CU stretched the window: round 5's "value in a VGPR came back wrong").  VS_BP_KNOB = 128 + 4096 n
    makes one wave of every workgroup sleep n x 512 cycles between the barrier and its read (n = 1, 2, 4, 16: inside and beyond the other
    waves' next round), on the quad walk and on the list walk (exact records, and head columns behind the pre-pass): results must equal
    the CSR scan's -- the round-5 code loses documents at every n, in a single process (profiles/r06_prune_decision_race.txt).  Small shards: every work item's first block, where all documents
    are candidates and a counter stands exactly at the limit after round one.  A subprocess: the library reads the knob once."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import numpy as np, torch, sys
sys.path.insert(0, %r)
import oracle
from vsearch_amd import _native as nat, synth
from vsearch_amd.device_index import DeviceIndex
for kind, walk, quant in ((0, -1, -1), (0, 0, 0), (2, -1, -1)):
    idx = DeviceIndex.synthetic(7, 0, 20_000 if kind == 0 else 40_000, 29523, 768, kind, 0, nat.VS_F32)
    for rep in range(3):
        q = torch.from_numpy(oracle.synth_queries(11 + rep, 32, 29523, 776, 0, 0, kind)).cuda()
        idx.set_option("blocked_postings", 0)
        ids0, sc0 = idx.search(q, 100)
        idx.set_option("blocked_postings", 1)
        idx.set_option("postings_walk", walk)
        idx.set_option("postings_quant", quant)
        ids1, sc1 = idx.search(q, 100)
        assert idx.info().last_path == 3, (kind, walk)
        assert (ids0.cpu() == ids1.cpu()).all() and (sc0.cpu() == sc1.cpu()).all(), (kind, walk, rep)
print("OK")
""" % repo
    for n in (1, 2, 4, 16):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VS_BP_KNOB=str(128 + 4096 * n)))
        assert r.returncode == 0 and "OK" in r.stdout, (n, r.stdout[-500:], r.stderr[-1500:])


def test_bot_index_row_sharded_two_shards_on_one_device_with_lock_step():
    """Two bag-of-token shards on ONE device, B >= 16, lock-step window on.  Shards on one device share one stream (api.hip
    `owns_stream`), so their walks run one after the other here; the bounded wait itself is exercised by
    test_lock_step_wait_times_out_instead_of_hanging.  Bit-equal to the unsharded index."""
    from vsearch_amd import _native as nat
    from vsearch_amd.device_index import DeviceIndex, ShardGroup
    n = 2 * 70_000
    whole = DeviceIndex.synthetic(5, 0, n, V, 86, synth.KIND_BOT, 0, nat.VS_NONE)
    halves = [DeviceIndex.synthetic(5, r0, n // 2, V, 86, synth.KIND_BOT, 0, nat.VS_NONE) for r0 in (0, n // 2)]
    for h in halves + [whole]:
        h.set_option("blocked_postings", 1)
    grp = ShardGroup(halves)
    q = torch.from_numpy(oracle.synth_queries(1, 64, V, 776, synth.VAL_DYADIC)).cuda()
    ids, sc = whole.search(q, 100)
    for _ in range(3):
        g_ids, g_sc = grp.search(q, 100)
        assert (g_ids.cpu() == ids.cpu()).all() and (g_sc.cpu() == sc.cpu()).all()
    assert halves[0].info().last_path == 3
    grp.close()
