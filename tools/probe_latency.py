"""Small-batch latency of Index.search through the C ABI: python tools/probe_latency.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
idx = DeviceIndex.synthetic(0, 0, N, 29523, 768, 0, 0, nat.VS_F32)
qs = DeviceIndex.synthetic(1, 0, 1024, 29523, 776, 0, 0, nat.VS_F32)
ip, ix, d = qs.export_csr()
q_all = torch.zeros((1024, 29523), device="cuda")
rows = torch.repeat_interleave(torch.arange(1024), torch.from_numpy(ip[1:] - ip[:-1]))
q_all[rows.cuda(), torch.from_numpy(ix).long().cuda()] = torch.from_numpy(d).cuda()
info = idx.info()
print(f"N={N} bytes/pass={info.bytes_per_pass/1e9:.2f} GB  ideal pass at 6.3 TB/s = {info.bytes_per_pass/6.3e9:.2f} ms")
for B in (1, 2, 4, 8, 16, 32, 64, 128):
    q = q_all[:B].contiguous()
    for _ in range(3):
        idx.search(q, 100)
    torch.cuda.synchronize()
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        ids, sc = idx.search(q, 100)
    t_host = (time.perf_counter() - t) / reps                 # what the host spends in the call (asynchronous when nothing forces a sync)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    Profile.enable(True); Profile.reset()
    idx.search(q, 100); torch.cuda.synchronize()
    ms, n = Profile.read("csr_scan_topk"); Profile.enable(False)
    print(f"B={B:4d}: {dt*1e3:8.3f} ms/call (host returns after {t_host*1e3:.3f} ms)  {B/dt:9.1f} q/s   scan kernel {ms:.3f} ms  path={idx.info().last_path}", flush=True)
