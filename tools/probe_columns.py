"""Uniform vs skewed column popularity (SURVEY 8(d) C3 secondary run) on the library's own generator:
python tools/probe_columns.py [N] [B] [HEAD]   -- (HEAD: option postings_head, see include/vsearch_hip.h) q/s of the postings filter search and of the 8-query CSR scan for both laws, checked
against each other bit for bit."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
HEAD = int(sys.argv[3]) if len(sys.argv) > 3 else -1
V = 29523
for name, kind in (("uniform", 0), ("zipf", 2)):
    idx = DeviceIndex.synthetic(0, 0, N, V, 768, kind, 0, nat.VS_F32)
    gen = DeviceIndex.synthetic(1, 0, B, V, 776, kind, 0, nat.VS_F32)
    ip, ix, d = gen.export_csr(); gen.close()
    q = torch.zeros((B, V), device="cuda")
    q[torch.from_numpy(np.repeat(np.arange(B), np.diff(ip))).cuda(), torch.from_numpy(ix).cuda()] = torch.from_numpy(d).cuda()
    idx.set_option("postings_head", HEAD)
    res = {}
    for mode in ("postings", "csr"):
        idx.set_option("blocked_postings", 1 if mode == "postings" else 0)
        idx.search(q, 100); torch.cuda.synchronize()
        Profile.enable(True); Profile.reset()
        reps = 2
        t = time.perf_counter()
        for _ in range(reps):
            ids, sc = idx.search(q, 100)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        ms, n = Profile.read("csr_scan_topk"); fb, _ = Profile.read("exact_fallback"); Profile.enable(False)
        inf = idx.info()
        res[mode] = (ids.cpu().numpy(), sc.cpu().numpy())
        print(json.dumps({"columns": name, "docs": N, "batch": B, "head": HEAD, "path": mode, "last_path": inf.last_path, "qps": B / dt, "ms_per_batch": dt * 1e3,
                          "scan_kernel_ms": ms / n, "fallback_ms": fb / reps, "fallback_queries": inf.last_fallbacks,
                          "walk_Gadds_per_s": (inf.last_walk_postings / (ms / n) / 1e6) if inf.last_path >= 2 else None}), flush=True)
    same = bool((res["postings"][0] == res["csr"][0]).all() and (res["postings"][1] == res["csr"][1]).all())
    print(json.dumps({"columns": name, "postings_equals_csr_bitwise": same}), flush=True)
    idx.close()
