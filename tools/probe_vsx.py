"""Native index file (.vsx) round trip: python tools/probe_vsx.py [N] -- save, load (VS_VERBOSE prints file -> pinned -> HBM GB/s), and
check that the loaded index answers like the original."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VS_VERBOSE"] = "1"
import numpy as np, torch
import oracle
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "probe_index.vsx")
idx = DeviceIndex.synthetic(0, 0, N, 29523, 768, 0, 0, nat.VS_F32)
q = torch.from_numpy(oracle.synth_queries(1, 16)).cuda()
ids0, sc0 = idx.search(q, 100)
t = time.perf_counter(); idx.save_native(path); t_save = time.perf_counter() - t
size = os.path.getsize(path)
print(f"saved {size / 1e9:.2f} GB in {t_save:.2f} s = {size / t_save / 1e9:.2f} GB/s (HBM -> host -> file)")
idx.close()
for rep in range(2):                                                  # the second load reads the page cache
    t = time.perf_counter(); idx2 = DeviceIndex.load_native(path); t_load = time.perf_counter() - t
    print(f"load {rep}: {size / 1e9:.2f} GB in {t_load:.2f} s = {size / t_load / 1e9:.2f} GB/s end to end (incl. payload validation)")
    ids1, sc1 = idx2.search(q, 100)
    assert (ids1 == ids0).all() and (sc1 == sc0).all()
    idx2.close()
os.remove(path)
print("vsx round trip ok")
