"""Filter-and-refine postings search vs the fp64 walk and the CSR scan: python tools/probe_filter.py [N] [B] [k] [store] [modes] [chunks] [lanes] [align]
Checks that the three paths return identical ids and bit-identical scores and prints their rates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
store = nat.VS_F16 if len(sys.argv) > 4 and sys.argv[4] == "fp16" else nat.VS_F32
modes = sys.argv[5].split(",") if len(sys.argv) > 5 else ["filter", "f64", "csr"]
chunk_list = [int(x) for x in sys.argv[6].split(",")] if len(sys.argv) > 6 else [0]
lanes = int(sys.argv[7]) if len(sys.argv) > 7 else 0
align = int(sys.argv[8]) if len(sys.argv) > 8 else -1
idx = DeviceIndex.synthetic(0, 0, N, 29523, 768, 0, 0, store)
q = torch.from_numpy(oracle.synth_queries(1, B)).cuda()
res = {}
idx.set_option("postings_lanes", lanes)
idx.set_option("postings_align", align)
idx.set_option("postings_walk", int(os.environ.get("VS_PROBE_WALK", "-1")))
idx.set_option("postings_arrange", int(os.environ.get("VS_PROBE_ARRANGE", "-1")))
if os.environ.get("VS_PROBE_ROWS"): idx.set_option("postings_rows", int(os.environ["VS_PROBE_ROWS"]))
for mode, chunks in [(m, c) for m in modes for c in (chunk_list if m != "csr" else [0])]:
    idx.set_option("postings_chunks", chunks)
    idx.set_option("blocked_postings", 0 if mode == "csr" else 1)
    idx.set_option("postings_filter", 1 if mode in ("filter", "filterx", "fallback", "fallbackx") else 0)
    idx.set_option("postings_quant", 0 if mode in ("filterx", "fallbackx") else -1)        # x = exact (fp32) records
    idx.set_option("postings_force_fallback", 1 if mode.startswith("fallback") else 0)
    torch.cuda.synchronize(); t = time.time()
    idx.search(q, k)
    torch.cuda.synchronize(); first = time.time() - t
    Profile.enable(True); Profile.reset()
    torch.cuda.synchronize(); t = time.time()
    reps = int(os.environ.get("VS_PROBE_REPS", "3"))
    for _ in range(reps):
        ids, sc = idx.search(q, k)
    torch.cuda.synchronize(); dt = (time.time() - t) / reps
    ms, n = Profile.read("csr_scan_topk"); rms, _ = Profile.read("refine_topk"); fms, _ = Profile.read("exact_fallback"); Profile.enable(False)
    res[mode] = (ids.cpu().numpy(), sc.cpu().numpy())
    inf = idx.info()
    print(f"{mode:7s} chunks={chunks} path={inf.last_path} first {first*1e3:.1f} ms, steady {dt*1e3:.2f} ms = {B/dt:.0f} q/s | walk {ms/reps:.2f} ms refine {rms/reps:.3f} ms "
          f"fallback {fms/reps:.3f} ms ({inf.last_fallbacks} queries) | {inf.last_walk_postings/(ms/reps)/1e6:.0f} Gadd/s", flush=True)
ref = modes[-1]
for mode in modes[:-1]:
    same_ids = (res[mode][0] == res[ref][0]).mean(); same_sc = (res[mode][1] == res[ref][1]).mean()
    print(f"{mode} vs {ref}: ids equal {same_ids:.6f}  scores bit-equal {same_sc:.6f}")
    if not (same_ids == 1.0 and same_sc == 1.0): print("MISMATCH", mode, np.abs(res[mode][1]-res[ref][1]).max())
