"""Quick perf probe of the CSR scan on the GPU box: python tools/probe_scan.py [N] [B] [k] [kind]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsearch_amd import _native as nat, synth
from vsearch_amd.device_index import DeviceIndex, Profile
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
kind = int(sys.argv[4]) if len(sys.argv) > 4 else 0
store = {0: nat.VS_F32, 1: nat.VS_NONE, 2: nat.VS_F16}[kind]
t = time.time()
idx = DeviceIndex.synthetic(0, 0, N, 29523, 86 if kind == 1 else 768, 1 if kind == 1 else 0, 0, store)
info = idx.info()
print(f"index N={N} built in {time.time()-t:.2f}s  bytes/pass={info.bytes_per_pass/1e9:.3f} GB G={info.lanes_per_row}", flush=True)
import torch
q = torch.from_numpy(oracle.synth_queries(1, B, val_law=1 if kind == 1 else 0)).cuda()
Profile.enable(True)
for it in range(3):
    Profile.reset()
    torch.cuda.synchronize(); t = time.time()
    ids, sc = idx.search(q, k)
    torch.cuda.synchronize(); dt = time.time() - t
    ms, n = Profile.read("csr_scan_topk")
    mms, mn = Profile.read("merge_topk")
    print(f"iter {it}: wall {dt*1e3:.2f} ms  {B/dt:.1f} q/s | scan {ms:.2f} ms ({n} launches) -> {B*info.bytes_per_pass/ms/1e6:.1f} GB/s "
          f"= {B*info.bytes_per_pass/ms/1e6/8000:.3f} of 8 TB/s | merge {mms:.3f} ms", flush=True)
