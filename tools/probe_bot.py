"""Binary bag-of-token index (C5 shape) option sweep: python tools/probe_bot.py [N] [B] -- q/s of the postings walk for
postings_chunks x postings_rows x postings_lanes, each checked against the first result."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 21_015_324
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rows_l = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
chunks_l = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
lanes_l = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
idx = DeviceIndex.synthetic(0, 0, N, 29523, 86, 1, 0, nat.VS_NONE)
q = torch.from_numpy(oracle.synth_queries(1, B, 29523, 776, 1)).cuda()
first = None
for rows, lanes, chunks in itertools.product(rows_l, lanes_l, chunks_l):
    idx.set_option("postings_rows", rows); idx.set_option("postings_lanes", lanes); idx.set_option("postings_walk", int(os.environ.get("VS_PROBE_WALK", "-1"))); idx.set_option("postings_chunks", chunks); idx.set_option("postings_packed", int(os.environ.get("VS_PROBE_PACKED", "-1")))
    idx.search(q, 100); torch.cuda.synchronize()
    Profile.enable(True); Profile.reset()
    t = time.perf_counter(); reps = 3
    for _ in range(reps):
        ids, sc = idx.search(q, 100)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    ms, n = Profile.read("csr_scan_topk"); Profile.enable(False)
    inf = idx.info()
    got = (ids.cpu().numpy(), sc.cpu().numpy())
    if first is None: first = got
    same = bool((got[0] == first[0]).all() and (got[1] == first[1]).all())
    print(f"rows={rows} lanes={lanes} chunks={chunks} path={inf.last_path}: {B/dt:.0f} q/s, walk {ms/n:.2f} ms, {inf.last_walk_postings/(ms/n)/1e6:.0f} Gadd/s, "
          f"copy {inf.aux_bytes/1e9:.2f} GB, fallbacks {inf.last_fallbacks}, same={same}", flush=True)
