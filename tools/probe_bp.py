"""Blocked-postings path vs the CSR multi-query scan: python tools/probe_bp.py [N] [B] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
idx = DeviceIndex.synthetic(0, 0, N, 29523, 768, 0, 0, nat.VS_F32)
q = torch.from_numpy(oracle.synth_queries(1, B)).cuda()
res = {}
for mode in (0, 1):
    idx.set_option("blocked_postings", mode)
    torch.cuda.synchronize(); t = time.time()
    idx.search(q, k)
    torch.cuda.synchronize(); first = time.time() - t
    Profile.enable(True); Profile.reset()
    torch.cuda.synchronize(); t = time.time()
    ids, sc = idx.search(q, k)
    torch.cuda.synchronize(); dt = time.time() - t
    ms, n = Profile.read("csr_scan_topk"); Profile.enable(False)
    res[mode] = (ids.cpu().numpy(), sc.cpu().numpy())
    inf = idx.info()
    print(f"   path={inf.last_path} scan bytes {inf.last_scan_bytes/1e9:.2f} GB -> {inf.last_scan_bytes/ms/1e6:.0f} GB/s; aux copy {inf.aux_bytes/1e9:.2f} GB")
    print(f"blocked_postings={mode}: first call {first*1e3:.1f} ms, steady {dt*1e3:.2f} ms = {B/dt:.0f} q/s (scan kernel {ms:.2f} ms)", flush=True)
same_ids = (res[0][0] == res[1][0]).mean(); same_sc = (res[0][1] == res[1][1]).mean()
print(f"ids equal: {same_ids:.6f}  scores bit-equal: {same_sc:.6f}")
