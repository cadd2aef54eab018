"""Quad chunks vs the record walk by row density: python tools/probe_quad_policy.py [N] [B]
For nnz per document in 64 .. 768: builds the synthetic index, times the filter search on quad chunks (postings_walk = 4) and on
records (postings_walk = 0), prints walk ms, the copies' bytes and the average list length of a 2048-document block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsearch_amd.device_index import DeviceIndex, Profile
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
V = 29523
q = torch.from_numpy(oracle.synth_queries(1, B)).cuda()
print(f"# {N} docs, {B} queries x 776 nnz, k = 100; list = nnz * 2048 / V postings per (block, column)")
print("nnz/doc  list  fill   quad ms  (copy GB)   records ms  (copy GB)   csr GB   auto picks")
for nnz in [int(x) for x in os.environ.get("VS_PROBE_NNZ", "64,128,192,256,384,512,768").split(",")]:
    idx = DeviceIndex.synthetic(0, 0, N, V, nnz, 0, 0, 0)
    idx.set_option("blocked_postings", 1)
    row = {}
    for name, walk in (("quad", 4), ("rec", 0), ("auto", -1)):
        idx.set_option("postings_walk", walk)
        ids, sc = idx.search(q, 100)
        torch.cuda.synchronize()
        Profile.enable(True); Profile.reset()
        for _ in range(3):
            ids, sc = idx.search(q, 100)
        torch.cuda.synchronize()
        ms, n = Profile.read("csr_scan_topk"); Profile.enable(False)
        inf = idx.info()
        row[name] = (ms / 3, inf.aux_bytes / 1e9, inf.postings_walk, inf.last_path, ids.cpu().numpy(), sc.cpu().numpy(), inf.device_bytes / 1e9)
    same = (row["quad"][4] == row["rec"][4]).all() and (row["quad"][5] == row["rec"][5]).all()
    lst = nnz * 2048 / V
    print(f"{nnz:7d} {lst:5.1f} {lst / 64:5.2f}  {row['quad'][0]:8.2f}  ({row['quad'][1]:6.2f})    {row['rec'][0]:8.2f}  ({row['rec'][1]:6.2f})   {row['auto'][6]:6.2f}   "
          f"walk {row['auto'][2]} path {row['auto'][3]}  {'bit-equal' if same else 'MISMATCH'}", flush=True)
    idx.close()
