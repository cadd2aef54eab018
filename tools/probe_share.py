"""What would a higher L2 hit rate buy the quad walk?  python tools/probe_share.py [N] [B]
The first-touch misses of a block are its lines (2 per column), whoever asks; the requests are 32 tiles x 12 416 per XCD.  Queries whose
columns all lie in the first V / f columns touch 1 / f of a block's lines with the SAME number of requests, lists of the same length, the
same arithmetic and LDS traffic: the miss rate falls by f (f = 2: what 16-query tiles would see) and nothing else changes.
(Round 6's first try -- identical tiles on one XCD -- changed nothing: the block's lines are all touched either way.)"""

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 21_015_324
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = 100
V = 29523
idx = DeviceIndex.synthetic(0, 0, N, V, 768, 0, 0, nat.VS_F32)
rng = np.random.default_rng(7)
def batch(f):
    q = np.zeros((B, V), dtype=np.float32)
    lim = V // f
    for i in range(B):
        c = rng.choice(lim, size=776, replace=False)
        q[i, c] = rng.random(776, dtype=np.float32) + 0.05
    return q
for f in [1, 2, 4, 8, 1]:
    q = torch.from_numpy(batch(f)).cuda()
    idx.search(q, k); torch.cuda.synchronize()
    Profile.enable(True); Profile.reset()
    reps = 4
    for _ in range(reps): idx.search(q, k)
    torch.cuda.synchronize()
    ms, n = Profile.read("csr_scan_topk"); Profile.enable(False)
    inf = idx.info()
    print(f"columns < V/{f}: path={inf.last_path} walk {ms/reps:.2f} ms  fallbacks {inf.last_fallbacks}", flush=True)
