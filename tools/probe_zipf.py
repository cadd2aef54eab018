"""Zipf-column corpus through the filter search (head pre-pass): python tools/probe_zipf.py [N] [B] -- for PMC passes (tools/pmc_walk.sh with
VS_PMC_PROBE=probe_zipf.py VS_PMC_ARGS=" ") and quick timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsearch_amd import synth
from vsearch_amd.device_index import DeviceIndex, Profile
import oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
idx = DeviceIndex.synthetic(0, 0, N, 29523, 768, synth.KIND_SKEW, 0, 0)
q = torch.from_numpy(oracle.synth_queries(1, B, kind=synth.KIND_SKEW)).cuda()
idx.search(q[:8], 100)
torch.cuda.synchronize()
Profile.enable(True); Profile.reset()
reps = int(os.environ.get("VS_PROBE_REPS", "2"))
t = time.time()
for _ in range(reps):
    ids, sc = idx.search(q, 100)
torch.cuda.synchronize(); dt = (time.time() - t) / reps
w, n = Profile.read("csr_scan_topk"); g, _ = Profile.read("head_gemm")
inf = idx.info()
print(f"zipf N={N} B={B}: {dt*1e3:.2f} ms/search = {B/dt:.0f} q/s | walk {w/reps:.2f} ms ({n/reps:.0f} launches) head gemm {g/reps:.2f} ms | heads {inf.head_columns} walk {inf.postings_walk} copy {inf.aux_bytes/1e9:.1f} GB fallbacks {inf.last_fallbacks} path {inf.last_path}")
