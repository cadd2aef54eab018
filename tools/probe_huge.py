"""Beyond 2^31 packets on one GPU (30 M docs x 768 nnz = 2.9e9 packets, 138 GB): top-k validated against the scores-only
kernel for both scan families, plus the tail rows checked against the host twin of the generator.
python tools/probe_huge.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vsearch_amd import _native as nat, synth
from vsearch_amd.device_index import DeviceIndex
from oracle import compare
import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
V = 29523
t = time.time()
idx = DeviceIndex.synthetic(0, 0, N, V, 768, 0, 0, nat.VS_F32)
info = idx.info()
print(f"N={N} packets={info.n_packets:.3e} (2^31={2**31:.3e}) bytes={info.device_bytes/1e9:.1f} GB built in {time.time()-t:.1f}s", flush=True)
q = oracle.synth_queries(1, 3)
# make the LAST rows the best hits of query 0 impossible to miss: not possible without editing the index; instead validate fully
allsc = idx.scores(q)
print("scores kernel done", allsc.shape, flush=True)
# tail rows against the host generator (exact same rows -> exact same dot products up to fp32 order)
ip, ix, d = oracle.synth_csr(0, N - 5, 5, V, 768)
for r in range(5):
    cols, vals = ix[ip[r]:ip[r + 1]], d[ip[r]:ip[r + 1]]
    want = (q[:, cols].astype(np.float64) * vals.astype(np.float64)).sum(1)
    got = allsc[:, N - 5 + r]
    assert np.allclose(got, want, rtol=2e-6), (r, got, want)
print("tail rows match the host generator", flush=True)
for qt in (0, 1):
    idx.set_queries_per_pass(qt)
    t = time.time()
    ids, sc = idx.search(q, 100)
    dt = time.time() - t
    compare.check_topk_valid(allsc, ids, sc, rtol=1e-4)
    hi = int((ids >= 2**31 // 96).sum())
    print(f"qt_pref={qt}: search {dt*1e3:.1f} ms, top-100 valid; {hi} of {ids.size} hits lie in rows whose packet index exceeds 2^31", flush=True)
print("huge index ok")
