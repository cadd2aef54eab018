#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE (/root/reference) in the build container.

Runs only where /root/reference is mounted (never on the GPU box). It imports the reference's own
modules (SURVEY.md §8(c) recipe B: third-party modules that are absent offline are stubbed in
sys.modules; the reference's source is imported from where it lies and never copied), drives them
on seeded synthetic inputs produced by ``vsearch_amd.synth`` and stores inputs-by-seed + outputs
verbatim as small ``.npz`` fixtures under ``tests/golden/``.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--only NAME]

Fixtures (SURVEY.md §8(c) list 1-7):
  sparse_utils.npz   elu1p / build_topk_mask / topk_sparsify      (src/ir/utils/sparse.py:6-19)
  bow_mask.npz       build_bow_mask, norm on/off                  (src/ir/utils/sparse.py:21-29)
  encoder_head.npz   LN -> vocab proj -> elu1p -> max-pool        (src/ir/encoder/vdr.py:71-75,83)
  embed_mask.npz     embed() mask logic on a fake forward         (src/ir/encoder/vdr.py:152-169)
  search_sparse_*.npz / search_dense.npz   Index.search           (src/ir/retriever/index.py:88-94)
  bot_build.npz      Retriever._build_bot_vectors                 (src/ir/retriever/retriever.py:208-253)
  search_bot.npz     BoTIndex.search, fp32 + dyadic queries
  retrieve.npz       Retriever.retrieve unbound (+ rerank)        (src/ir/retriever/retriever.py:107-148)
  save_load.npz      SparseIndex.save / load, shift, 2 shards     (src/ir/retriever/index.py:163-202)
"""
import argparse
import json
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

V = 29523
VOCAB = 30522
SHIFT = 999


def import_reference():
    import torch  # noqa: F401
    from transformers import (PreTrainedModel, PretrainedConfig, AutoModel, AutoTokenizer,  # noqa: F401
                              BertConfig, BatchEncoding)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    stub("wordcloud", WordCloud=_Dummy)
    stub("spacy")
    stub("pynvml")
    stub("hydra")
    stub("hydra.utils", instantiate=lambda *a, **k: None)
    stub("omegaconf", DictConfig=dict, OmegaConf=_Dummy)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms", Compose=_Dummy, CenterCrop=_Dummy,
                         Normalize=_Dummy, Resize=_Dummy, ToTensor=_Dummy)
    sys.path.insert(0, REF)
    from src.ir import Retriever
    from src.ir.retriever import index as ref_index
    from src.ir.utils import sparse as ref_sparse
    from src.ir.retriever.index_utils import get_first_unique_n
    return types.SimpleNamespace(Retriever=Retriever, index=ref_index, sparse=ref_sparse,
                                 get_first_unique_n=get_first_unique_n)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def torch_csr(indptr, indices, data, shape):
    import torch
    return torch.sparse_csr_tensor(torch.from_numpy(indptr), torch.from_numpy(indices.astype(np.int64)),
                                   torch.from_numpy(data), size=shape)


# ----------------------------------------------------------------------------------------------
def g_sparse_utils(ref):
    import torch
    from vsearch_amd import synth
    x_small = synth.dense_uniform(11, (4, 257), -6.0, 6.0)
    elu = ref.sparse.elu1p(torch.from_numpy(x_small)).numpy()
    seed, rows = 12, 8
    x = synth.dense_tiefree(seed, (rows, V))
    out = {}
    for k in (1, 100, 768):
        mask = ref.sparse.build_topk_mask(torch.from_numpy(x), k=k).numpy()
        assert mask.sum(1).tolist() == [k] * rows
        out[f"mask_k{k}"] = np.packbits(mask, axis=1)
    mask_np = ref.sparse.build_topk_mask(x, k=768).numpy()      # numpy-input branch (sparse.py:9-10)
    assert (np.packbits(mask_np, axis=1) == out["mask_k768"]).all()
    sp = ref.sparse.topk_sparsify(torch.from_numpy(x), 768).numpy()
    nz_r, nz_c = np.nonzero(sp)
    save("sparse_utils", elu_in=x_small, elu_out=elu, x_seed=np.int64(seed), x_rows=np.int64(rows),
         sparsify_cols=nz_c.reshape(rows, 768).astype(np.int32), sparsify_vals=sp[nz_r, nz_c].reshape(rows, 768),
         **out)


def bow_id_batches():
    rng = np.random.default_rng(5)
    b0 = np.array([[101, 2054, 2003, 2054, 102, 0, 0, 0],
                   [101, 999, 1000, 30521, 102, 0, 0, 0],
                   [101, 998, 999, 999, 999, 5000, 102, 0],
                   [0, 0, 0, 0, 0, 0, 0, 0]], dtype=np.int64)
    b1 = rng.integers(0, VOCAB, size=(6, 64)).astype(np.int64)
    b1[:, 0] = 101
    b1[2, 10:] = 0
    return [b0, b1]


def g_bow_mask(ref):
    import torch
    arrays = {}
    for i, ids in enumerate(bow_id_batches()):
        for norm in (False, True):
            m = ref.sparse.build_bow_mask(torch.from_numpy(ids), vocab_size=VOCAB, shift_num=SHIFT, norm=norm).numpy()
            assert m.shape == (ids.shape[0], V) and m.dtype == np.float32
            r, c = np.nonzero(m)
            tag = f"b{i}_{'norm' if norm else 'raw'}"
            arrays[f"{tag}_rows"] = r.astype(np.int32)
            arrays[f"{tag}_cols"] = c.astype(np.int32)
            arrays[f"{tag}_vals"] = m[r, c]
        arrays[f"b{i}_ids"] = ids
        m0 = ref.sparse.build_bow_mask(torch.from_numpy(ids), vocab_size=VOCAB, shift_num=0).numpy()
        arrays[f"b{i}_noshift_nnz"] = m0.sum(1).astype(np.int64)
    save("bow_mask", **arrays)


def g_encoder_head(ref):
    """vdr.py:71-75,83 restated call-for-call on synthetic hidden states (BERT weights are not
    available offline): ln -> @ W[shift:].t() -> elu1p -> max(1)[0] -> (optional) F.normalize."""
    import torch
    import torch.nn.functional as F
    from vsearch_amd import synth
    B, L, H, vocab, shift = 4, 16, 768, 3000, 999
    hidden = synth.dense_uniform(21, (B, L, H), -2.0, 2.0)
    W = synth.dense_uniform(22, (vocab, H), -0.08, 0.08)
    ln_w = synth.dense_uniform(23, (H,), 0.5, 1.5)
    ln_b = synth.dense_uniform(24, (H,), -0.1, 0.1)
    ln = torch.nn.LayerNorm(H)
    with torch.no_grad():
        ln.weight.copy_(torch.from_numpy(ln_w))
        ln.bias.copy_(torch.from_numpy(ln_b))
        h_ln = ln(torch.from_numpy(hidden))
        vocab_embs = h_ln @ torch.from_numpy(W)[shift:, :].t()
        pre_max = vocab_embs.max(1)[0]                      # for the elu1p∘max commute check
        vocab_embs = ref.sparse.elu1p(vocab_embs)
        emb = vocab_embs.max(1)[0]
        emb_norm = F.normalize(emb)
    assert torch.allclose(ref.sparse.elu1p(pre_max), emb, rtol=0, atol=0)
    save("encoder_head", shape=np.array([B, L, H, vocab, shift]), seeds=np.array([21, 22, 23, 24]),
         emb=emb.numpy(), emb_norm=emb_norm.numpy(), logits_max=pre_max.numpy())


def g_embed_mask(ref):
    """VDREncoder.embed mask logic (vdr.py:152-169) driven with the class's own method on a fake
    self whose forward returns a seeded dense embedding."""
    import torch
    from functools import partial
    from vsearch_amd import synth
    from src.ir.encoder.vdr import VDREncoder
    ids = bow_id_batches()[1]
    B = ids.shape[0]
    dense = synth.dense_tiefree(31, (B, V), 0.05, 6.0)

    class Enc(dict):
        """BatchEncoding stand-in: mapping (for ``self(**encoding)``) with an ``input_ids`` attribute."""

    class FakeEnc:
        training = False
        config = types.SimpleNamespace(max_len=64, topk=768)
        build_bow_mask = staticmethod(partial(ref.sparse.build_bow_mask, vocab_size=VOCAB, shift_num=SHIFT, norm=False))

        def encode(self, texts, max_len=None):
            sel = [int(t) for t in texts]
            e = Enc(sel=sel)
            e.input_ids = torch.from_numpy(ids[sel])
            return e

        def __call__(self, sel):
            return torch.from_numpy(dense[sel].copy())

        def eval(self):
            pass

    fake = FakeEnc()
    texts = [str(i) for i in range(B)]
    arrays = {"ids": ids, "dense_seed": np.int64(31)}
    for name, kw in {
        "top768_lex": dict(topk=768, activate_lexical=True),
        "top768_nolex": dict(topk=768, activate_lexical=False),
        "top0_lex": dict(topk=0, activate_lexical=True),
        "all": dict(topk=-1, activate_lexical=False),
        "bow": dict(bow=True),
        "top16_lex_bs4": dict(topk=16, activate_lexical=True, batch_size=4),
    }.items():
        out = VDREncoder.embed(fake, texts, **kw).numpy()
        r, c = np.nonzero(out)
        arrays[f"{name}_rows"] = r.astype(np.int32)
        arrays[f"{name}_cols"] = c.astype(np.int32)
        if name != "all":
            arrays[f"{name}_vals"] = out[r, c]
        else:
            assert (out == dense).all()
    save("embed_mask", **arrays)


def search_case(ref, cls, vector, q, k):
    import torch
    idx = cls.__new__(cls)
    idx.device = "cpu"
    idx.vector = vector
    idx.data = None
    res = idx.search(torch.from_numpy(q), k)
    assert isinstance(res, ref.index.SearchResults)
    return res.ids.numpy(), res.scores.numpy()


def g_search_sparse(ref):
    from vsearch_amd import synth
    cases = [  # name, index seed, N, query seed, B, ks
        ("search_sparse_n2000", 0, 2000, 1, 8, (1, 100, 2000)),
        ("search_sparse_n20000", 0, 20000, 1, 32, (100,)),
    ]
    for name, iseed, n, qseed, b, ks in cases:
        indptr, indices, data = synth.synth_csr(iseed, 0, n)
        vec = torch_csr(indptr, indices, data, (n, V))
        q = synth.synth_queries(qseed, b)
        arrays = dict(index_seed=np.int64(iseed), n=np.int64(n), query_seed=np.int64(qseed), b=np.int64(b),
                      nnz_row=np.int64(768), nnz_q=np.int64(776))
        for k in ks:
            ids, scores = search_case(ref, ref.index.SparseIndex, vec, q, k)
            arrays[f"ids_k{k}"] = ids.astype(np.int32)
            arrays[f"scores_k{k}"] = scores
        # exact (fp64) scores of the reference's picks, for the tolerance statement in tests
        save(name, **arrays)


def g_search_dense(ref):
    import torch
    from vsearch_amd import synth
    n, b = 2000, 8
    indptr, indices, data = synth.synth_csr(0, 0, n)
    dense = torch_csr(indptr, indices, data, (n, V)).to_dense()
    q = synth.synth_queries(1, b)
    arrays = dict(index_seed=np.int64(0), n=np.int64(n), query_seed=np.int64(1), b=np.int64(b))
    for k in (1, 100):
        ids, scores = search_case(ref, ref.index.Index, dense, q, k)
        arrays[f"ids_k{k}"] = ids.astype(np.int32)
        arrays[f"scores_k{k}"] = scores
    # a genuinely dense index (no zeros): seeded uniform rows
    n2 = 512
    dense2 = synth.dense_uniform(41, (n2, V), 0.0, 1.0)
    q2 = synth.dense_uniform(42, (4, V), 0.0, 1.0)
    ids, scores = search_case(ref, ref.index.Index, torch.from_numpy(dense2), q2, 50)
    arrays["full_ids_k50"] = ids.astype(np.int32)
    arrays["full_scores_k50"] = scores
    arrays["full_shape"] = np.array([n2, 4, 41, 42])
    try:
        search_case(ref, ref.index.Index, dense, q, n + 1)
        arrays["k_gt_n_raises"] = np.bool_(False)
    except RuntimeError:
        arrays["k_gt_n_raises"] = np.bool_(True)
    save("search_dense", **arrays)


class FakeTokenizer:
    """Texts are space-separated token ids; mimics HF call signature used at retriever.py:238."""
    vocab = range(VOCAB)

    def __call__(self, texts, max_length=None, truncation=False):
        out = []
        for t in texts:
            ids = [int(x) for x in t.split()]
            if truncation and max_length is not None and len(ids) > max_length:
                ids = ids[:max_length - 1] + [102]
            out.append(ids)
        return {"input_ids": out}


def bot_texts(n, seed, lo=8, hi=160):
    rng = np.random.default_rng(seed)
    texts = []
    for i in range(n):
        ln = int(rng.integers(lo, hi))
        # zipf-ish: repeated tokens inside a doc, a few ids below the shift
        body = rng.integers(1996, 12000, size=ln)
        body[rng.random(ln) < 0.15] = rng.integers(0, 1100)
        body[rng.random(ln) < 0.3] = body[0]
        texts.append(" ".join(map(str, [101] + body.tolist() + [102])))
    return texts


def g_bot_build(ref):
    import torch  # noqa: F401
    texts = bot_texts(40, 7)
    fake = types.SimpleNamespace(encoder_p=types.SimpleNamespace(tokenizer=FakeTokenizer()))
    arrays = {"texts": np.array(texts)}
    for tag, kw in {"full": {}, "max16": {"max_token": 16}, "len32": {"max_len": 32}, "fp32": {"fp16": False}}.items():
        csr = ref.Retriever._build_bot_vectors(fake, texts, batch_size=64, **kw)
        assert csr.layout == __import__("torch").sparse_csr
        arrays[f"{tag}_indptr"] = csr.crow_indices().numpy()
        arrays[f"{tag}_indices"] = csr.col_indices().numpy().astype(np.int32)
        arrays[f"{tag}_dtype"] = np.array(str(csr.values().dtype))
        assert (csr.values().float() == 1).all()
        arrays[f"{tag}_shape"] = np.array(csr.shape)
    save("bot_build", **arrays)


def g_search_bot(ref):
    from vsearch_amd import synth
    n, b = 5000, 16
    indptr, indices, data = synth.synth_csr(3, 0, n, nnz=86, kind=synth.KIND_BOT)
    vec = torch_csr(indptr, indices, data, (n, V))          # fp32-cast BoT (CPU has no fp16 CSR matmul)
    arrays = dict(index_seed=np.int64(3), n=np.int64(n), b=np.int64(b), nnz=np.int64(86))
    q_f = synth.synth_queries(4, b)
    q_d = synth.synth_queries(5, b, val_law=synth.VAL_DYADIC)
    for tag, q in (("f32", q_f), ("dyadic", q_d)):
        for k in (10, 100):
            ids, scores = search_case(ref, ref.index.BoTIndex, vec, q, k)
            arrays[f"{tag}_ids_k{k}"] = ids.astype(np.int32)
            arrays[f"{tag}_scores_k{k}"] = scores
    arrays["query_seeds"] = np.array([4, 5])
    save("search_bot", **arrays)


def g_retrieve(ref):
    import torch
    from functools import partial
    from vsearch_amd import synth
    n, b, k = 600, 4, 10
    indptr, indices, data = synth.synth_csr(8, 0, n, nnz=86, kind=synth.KIND_BOT)
    bot = ref.index.BoTIndex.__new__(ref.index.BoTIndex)
    bot.device, bot.data, bot.low_memory = "cpu", [f"{i}" for i in range(n)], False
    bot.vector = torch_csr(indptr, indices, data, (n, V))
    ip2, ix2, d2 = synth.synth_csr(9, 0, n)                  # "parametric" passage embeddings for rerank
    p_dense = torch_csr(ip2, ix2, d2, (n, V)).to_dense()

    def fake_embed(texts, batch_size=32, require_grad=False, **kw):
        return p_dense[[int(t) for t in texts]]

    fake = types.SimpleNamespace(index=bot, device="cpu",
                                 encoder_q=types.SimpleNamespace(config=types.SimpleNamespace(topk=768)),
                                 encoder_p=types.SimpleNamespace(embed=fake_embed))
    fake.process_query = partial(ref.Retriever.process_query, fake)
    q = synth.synth_queries(10, b)
    arrays = dict(n=np.int64(n), b=np.int64(b), k=np.int64(k), seeds=np.array([8, 9, 10]))
    r_t = ref.Retriever.retrieve(fake, torch.from_numpy(q), k=k)
    r_n = ref.Retriever.retrieve(fake, q, k=k)
    assert (r_t.ids == r_n.ids).all()
    arrays["ids"], arrays["scores"] = r_t.ids.numpy().astype(np.int32), r_t.scores.numpy()
    r_r = ref.Retriever.retrieve(fake, torch.from_numpy(q), k=k, rerank=True)
    arrays["rerank_ids"], arrays["rerank_scores"] = r_r.ids.numpy().astype(np.int32), r_r.scores.numpy()
    # sparse (non-BoT) index: rerank flag must be ignored (retriever.py:137)
    sp = ref.index.SparseIndex.__new__(ref.index.SparseIndex)
    sp.device, sp.data, sp.low_memory, sp.vector = "cpu", bot.data, False, torch_csr(ip2, ix2, d2, (n, V))
    fake.index = sp
    r_s = ref.Retriever.retrieve(fake, torch.from_numpy(q), k=k, rerank=True)
    arrays["sparse_ids"], arrays["sparse_scores"] = r_s.ids.numpy().astype(np.int32), r_s.scores.numpy()
    try:
        ref.Retriever.process_query(fake, 3.14)
        arrays["bad_query_raises"] = np.bool_(False)
    except NotImplementedError:
        arrays["bad_query_raises"] = np.bool_(True)
    save("retrieve", **arrays)


def g_save_load(ref):
    import torch  # noqa: F401
    from scipy.sparse import load_npz
    from vsearch_amd import synth
    arrays = {}
    with tempfile.TemporaryDirectory() as td:
        n = 10
        ip, ix, d = synth.synth_csr(13, 0, 2 * n)
        for s in range(2):
            sl = slice(ip[s * n], ip[(s + 1) * n])
            idx = ref.index.SparseIndex.__new__(ref.index.SparseIndex)
            idx.device, idx.data = "cpu", None
            idx.vector = torch_csr(ip[s * n:(s + 1) * n + 1] - ip[s * n], ix[sl], d[sl], (n, V))
            idx.save(os.path.join(td, f"index{s}.npz"))
        with np.load(os.path.join(td, "index0.npz")) as z:
            arrays["manifest_keys"] = np.array(sorted(z.files))
            arrays["manifest_dtypes"] = np.array([str(z[k].dtype) for k in sorted(z.files)])
            arrays["manifest_format"] = z["format"]
            arrays["manifest_shape"] = z["shape"]
        m = load_npz(os.path.join(td, "index0.npz"))
        arrays["reload_indptr"], arrays["reload_indices"], arrays["reload_data"] = m.indptr, m.indices, m.data
        for tag, kw in {"shift0": dict(shift=0), "shift999": dict(shift=999)}.items():
            li = ref.index.SparseIndex(os.path.join(td, "index*.npz"), None, fp16=False, device="cpu", **kw)
            arrays[f"{tag}_shape"] = np.array(li.vector.shape)
            arrays[f"{tag}_indptr"] = li.vector.crow_indices().numpy()
            arrays[f"{tag}_indices"] = li.vector.col_indices().numpy().astype(np.int32)
            arrays[f"{tag}_data"] = li.vector.values().numpy()
            arrays[f"{tag}_str"] = np.array(str(li))
    arrays["seed"] = np.int64(13)
    save("save_load", **arrays)


GENERATORS = {
    "sparse_utils": g_sparse_utils, "bow_mask": g_bow_mask, "encoder_head": g_encoder_head,
    "embed_mask": g_embed_mask, "search_sparse": g_search_sparse, "search_dense": g_search_dense,
    "bot_build": g_bot_build, "search_bot": g_search_bot, "retrieve": g_retrieve, "save_load": g_save_load,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    import torch
    import scipy
    for name, fn in GENERATORS.items():
        if args.only and args.only != name:
            continue
        print(f"[{name}]")
        fn(ref)
    meta = {"torch": torch.__version__, "numpy": np.__version__, "scipy": scipy.__version__,
            "reference": "jzhoubu/vsearch @ 2024-12-18 (/root/reference)", "threads": torch.get_num_threads(),
            "note": "ids/scores are the reference's own outputs on this container's CPU (MKL sparse addmm + topk); "
                    "tie order and fp32 summation order are implementation-defined -> compare with oracle.compare"}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
