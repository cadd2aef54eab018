#!/usr/bin/env python3
"""Times the fused mask stage -> CSR (vs_embed_mask_to_csr) on a [B, V] fp32 batch and checks it against torch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vsearch_amd.ir.utils import sparse as sp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
V = int(sys.argv[2]) if len(sys.argv) > 2 else 29523
K = int(sys.argv[3]) if len(sys.argv) > 3 else 768
LEX = int(sys.argv[4]) if len(sys.argv) > 4 else 1
VOC, SHIFT, L = V + 999, 999, 128
g = torch.Generator(device="cuda").manual_seed(0)
emb = torch.rand((B, V), device="cuda", generator=g) * 3
tok = torch.randint(SHIFT, VOC, (B, L), device="cuda", generator=g)
rp, cols, vals = sp.embed_mask_to_csr(emb, tok, VOC, SHIFT, K, bool(LEX))
mask = torch.zeros_like(emb, dtype=torch.bool)
mask.scatter_(1, emb.topk(K, dim=1).indices, True)
if LEX: mask.scatter_(1, tok - SHIFT, True)
want = (emb * mask).to_sparse_csr()
ok = bool((want.crow_indices() == rp).all().item()) and bool((want.col_indices() == cols).all().item()) and bool((want.values() == vals).all().item())
print("equal to torch:", ok, "nnz/row", float(rp[-1].item()) / B)
for _ in range(3): sp.embed_mask_to_csr(emb, tok, VOC, SHIFT, K, bool(LEX))
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); st.record()
for _ in range(50): sp.embed_mask_to_csr(emb, tok, VOC, SHIFT, K, bool(LEX))
en.record(); torch.cuda.synchronize()
ms = st.elapsed_time(en) / 50
print(f"mask -> CSR B={B} V={V} k={K} lexical={LEX}: {ms:.4f} ms per call")
from vsearch_amd import _native as nat
try:
    from vsearch_amd.device_index import Profile
    Profile.enable(True); Profile.reset()
    for _ in range(50): sp.embed_mask_to_csr(emb, tok, VOC, SHIFT, K, bool(LEX))
    torch.cuda.synchronize(); Profile.enable(False)
    tot, n = Profile.read("mask_to_csr")
    print(f"kernel time (hipEvents around the launches): {tot / 50:.4f} ms")
except Exception as e:
    print("no profile scope:", e)
