"""Randomised parity hunt: random corpus shape / column law / store / batch / k / layout options, the filter search against the
8-query CSR scan of the same index, bit for bit.  python tools/stress_random.py [iters] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsearch_amd import _native as nat
from vsearch_amd import synth
from vsearch_amd.device_index import DeviceIndex
import oracle

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
V = 29523
bad = 0
for it in range(iters):
    kind = int(rng.choice([synth.KIND_VDR, synth.KIND_BOT, synth.KIND_SKEW]))
    store = nat.VS_NONE if kind == 1 else int(rng.choice([nat.VS_F32, nat.VS_F16]))
    n = int(rng.choice([5000, 9000, 30000, 70000, 150000]))
    nnz = int(rng.choice([86, 86, 400])) if kind == 1 else int(rng.choice([300, 768]))
    B = int(rng.choice([1, 3, 8, 9, 40, 129, 300]))
    k = int(rng.choice([1, 10, 100, 300]))
    opts = dict(postings_align=int(rng.choice([0, 1])), postings_lanes=int(rng.choice([0, 4, 8])), postings_rows=int(rng.choice([0, 512, 1024, 1920])),
                postings_chunks=int(rng.choice([0, 1, 3])), postings_head=int(rng.choice([-1, 0, 2, 16])), postings_quant=int(rng.choice([-1, 0])),
                postings_walk=int(rng.choice([-1, 0, 4, 4, -1])), postings_pace=int(rng.choice([-1, 0, 4])), postings_arrange=int(rng.choice([0, 1])),
                postings_head_product=int(rng.choice([-1, 0, 1])), postings_packed=int(rng.choice([-1, 0])))
    if n < 66000 and kind == 1:
        n = 70000                                                     # (the binary index takes the postings walk from 65 536 documents)
    idx = DeviceIndex.synthetic(it, 0, n, V, nnz, kind, 0, store)
    q = oracle.synth_queries(100 + it, B, V, 776 if kind != 1 else 776, synth.VAL_DYADIC if kind == 1 and rng.random() < 0.5 else 0,
                             **({"kind": synth.KIND_SKEW} if kind == 2 else {}))
    qd = torch.from_numpy(q).cuda()
    idx.set_option("blocked_postings", 0)
    ref_ids, ref_sc = idx.search(qd, k)
    ref_ids, ref_sc = ref_ids.cpu().numpy(), ref_sc.cpu().numpy()
    idx.set_option("blocked_postings", 1)
    for name, value in opts.items():
        idx.set_option(name, value)
    ids, sc = idx.search(qd, k)
    info = idx.info()
    same = bool((ids.cpu().numpy() == ref_ids).all() and (sc.cpu().numpy() == ref_sc).all())
    print(f"iter {it}: kind {kind} store {store} n {n} nnz {nnz} B {B} k {k} {opts} -> path {info.last_path} heads {info.head_columns} "
          f"fallbacks {info.last_fallbacks} {'ok' if same else 'MISMATCH'}", flush=True)
    bad += 0 if same else 1
    idx.close()
print("stress_random:", "ok" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
