#!/usr/bin/env python3
"""ONE secondary leg of bench.py on its own (for rocprofv3 --pmc passes): python3 tools/leg_pmc.py C3|C5|zipf|fp16 [searches]
Builds the leg's synthetic index, one 8-query warm-up search (builds the postings copy), then `searches` (default 1) full 1024-query
searches.  Prints the leg's record (no oracle parity: bench.py does that)."""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # run_leg does one warm-up search + `steps` timed ones
leg = {n: (key, a) for n, key, a in bench.SECONDARY_LEGS}[name]
from vsearch_amd import _native as nat
nat.require_device()
torch.cuda.set_device(0)
key, a = leg
rec = bench.run_leg(key, a[0], a[1], a[2], a[3], a[4], max(steps, 1) if steps else 1, bench.BATCH, bench.K, 0, torch.device("cuda", 0), parity=False, **{k: v for k, v in a[6].items() if k != "exact"})
print(json.dumps({key: rec}))
