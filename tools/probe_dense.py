"""Perf probe of the dense index path (config C2): python tools/probe_dense.py [N] [B] [k]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vsearch_amd.device_index import DeviceIndex, Profile
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
V = 29523
torch.manual_seed(0)
g = torch.Generator(device="cuda").manual_seed(0)
# VDR-like rows: 768 positive values at random columns, zeros elsewhere
mat = torch.zeros((N, V), device="cuda")
cols = torch.rand((N, V), device="cuda", generator=g).topk(768, dim=1).indices if N <= 20000 else None
if cols is None:
    for s in range(0, N, 10000):
        c = torch.rand((min(10000, N - s), V), device="cuda", generator=g).topk(768, dim=1).indices
        mat[s:s + c.shape[0]].scatter_(1, c, 0.01 + 3 * torch.rand(c.shape, device="cuda", generator=g))
else:
    mat.scatter_(1, cols, 0.01 + 3 * torch.rand(cols.shape, device="cuda", generator=g))
q = torch.zeros((B, V), device="cuda")
qc = torch.rand((B, V), device="cuda", generator=g).topk(776, dim=1).indices
q.scatter_(1, qc, 0.01 + 3 * torch.rand(qc.shape, device="cuda", generator=g))
idx = DeviceIndex.from_dense(mat)
Profile.enable(True)
for it in range(3):
    Profile.reset()
    torch.cuda.synchronize(); t = time.time()
    ids, sc = idx.search(q, k)
    torch.cuda.synchronize(); dt = time.time() - t
    ms, n = Profile.read("dense_scores"); mms, _ = Profile.read("merge_topk")
    print(f"iter {it}: wall {dt*1e3:.1f} ms {B/dt:.0f} q/s | gemm {ms:.2f} ms = {2*B*V*N/ms/1e9:.1f} TF/s | merge {mms:.2f} ms", flush=True)
t = time.time(); ref = (q @ mat.t()).topk(k); torch.cuda.synchronize(); t = time.time()
ref = (q @ mat.t()).topk(k); torch.cuda.synchronize(); print(f"torch matmul+topk: {(time.time()-t)*1e3:.1f} ms")
print("ids equal frac:", (ref.indices == ids).float().mean().item(), "max rel err", ((ref.values - sc).abs() / ref.values).max().item())
