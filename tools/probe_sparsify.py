"""Sparsify-stage kernels vs the reference's torch ops on the same GPU: python tools/probe_sparsify.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vsearch_amd.ir.utils import sparse as sp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
V, VOC, SHIFT, L = 29523, 30522, 999, 128
g = torch.Generator(device="cuda").manual_seed(0)
emb = torch.rand((B, V), device="cuda", generator=g) * 3
ids = torch.randint(999, VOC, (B, L), device="cuda", generator=g)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def ref_topk_mask(e, k=768):
    idx = torch.topk(e, k, dim=-1).indices
    return torch.zeros_like(e, dtype=torch.bool).scatter_(-1, idx, True)


def ref_bow(i):
    m = torch.zeros((i.shape[0], VOC), device=i.device).scatter_(-1, i, 1.0)
    return m[:, SHIFT:]


def ref_embed(e, i):
    e = e.clone()
    mask = ref_topk_mask(e) | ref_bow(i).bool()
    return e * mask


print(f"B={B}  [B,V] fp32 = {B*V*4/1e6:.0f} MB")
print(f"build_topk_mask   : hip {t(lambda: sp.build_topk_mask(emb, 768)):7.3f} ms   torch {t(lambda: ref_topk_mask(emb)):7.3f} ms")
print(f"build_bow_mask    : hip {t(lambda: sp.build_bow_mask(ids, VOC, SHIFT)):7.3f} ms   torch {t(lambda: ref_bow(ids)):7.3f} ms")
e2 = emb.clone()
print(f"embed mask (in place, topk|lexical): hip {t(lambda: sp.apply_embed_mask_(e2, ids, VOC, SHIFT, 768, True)):7.3f} ms   torch {t(lambda: ref_embed(emb, ids)):7.3f} ms")
print(f"elu1p             : hip {t(lambda: sp.elu1p(emb)):7.3f} ms   torch {t(lambda: F.elu(emb) + 1):7.3f} ms")
sparse = sp.topk_sparsify(emb, 768)
print(f"dense_to_csr      : hip {t(lambda: sp.dense_to_csr(sparse)):7.3f} ms   torch {t(lambda: sparse.to_sparse_csr()):7.3f} ms")
