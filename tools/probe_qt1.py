import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile
import oracle
N = int(sys.argv[1]); B = int(sys.argv[2])
idx = DeviceIndex.synthetic(0, 0, N, 29523, 768, 0, 0, nat.VS_F32)
q = torch.from_numpy(oracle.synth_queries(1, B)).cuda()
idx.set_queries_per_pass(1)
info = idx.info()
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    ids, sc = idx.search(q, 100)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"Qt=1 N={N} B={B}: {dt*1e3:.1f} ms  {B/dt:.1f} q/s  -> {B*info.bytes_per_pass/dt/1e12:.2f} TB/s algorithmic", flush=True)
# dense queries (all dims non-zero) take the same path by necessity
qd = (torch.rand((min(B, 64), 29523), device="cuda") + 0.01)
idx.set_queries_per_pass(0)
torch.cuda.synchronize(); t = time.time(); ids, sc = idx.search(qd, 100); torch.cuda.synchronize(); dt = time.time() - t
print(f"dense queries B={qd.shape[0]} (qt used {idx.info().queries_per_pass}): {dt*1e3:.1f} ms {qd.shape[0]/dt:.1f} q/s")
