"""Encoder head: fused vs torch GEMM + pooling. python tools/probe_head.py [B] [L]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vsearch_amd.ir.utils import sparse as sp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H, V = 768, 29523
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn((B, L, H), device="cuda", generator=g)
W = torch.randn((V, H), device="cuda", generator=g) * 0.05
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3, out
ms_f, a = t(lambda: sp.head_project_pool(h, W))
ms_u, b = t(lambda: sp.head_pool(h @ W.t()))
ms_t, c = t(lambda: (torch.nn.functional.elu(h @ W.t()) + 1).max(1)[0])
flops = 2.0 * B * L * H * V
print(f"B={B} L={L}: fused {ms_f:.2f} ms ({flops/ms_f/1e9:.1f} TF/s) | torch GEMM + vs_head_pool {ms_u:.2f} ms | all-torch (reference expression) {ms_t:.2f} ms")
print("max abs diff fused vs torch:", (a - c).abs().max().item(), " peak mem of logits avoided: %.2f GB" % (B * L * V * 4 / 1e9))
