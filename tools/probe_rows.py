"""Valued index: walk time against postings_rows / postings_lanes / postings_align: python tools/probe_rows.py [N] [rows,...] [lanes,...] [align,...]"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rows_l = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
lanes_l = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
align_l = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [-1]
idx = DeviceIndex.synthetic(0, 0, N, 29523, 768, 0, 0, nat.VS_F32)
q = torch.from_numpy(oracle.synth_queries(1, 1024)).cuda()
for rows, lanes, align in itertools.product(rows_l, lanes_l, align_l):
    idx.set_option("postings_rows", rows); idx.set_option("postings_lanes", lanes); idx.set_option("postings_align", align)
    idx.search(q, 100); torch.cuda.synchronize()
    Profile.enable(True); Profile.reset()
    for _ in range(3):
        idx.search(q, 100)
    torch.cuda.synchronize()
    ms, n = Profile.read("csr_scan_topk"); Profile.enable(False)
    inf = idx.info()
    print(f"rows={rows} lanes={lanes} align={align}: walk {ms / n:.2f} ms, {inf.last_walk_postings / (ms / n) / 1e6:.0f} Gadd/s, copy {inf.aux_bytes / 1e9:.2f} GB, fallbacks {inf.last_fallbacks}", flush=True)
