"""Randomised stress of the dense index (fp32 MFMA search and the sparsity-aware CSR route) against a float64 torch
matmul: python tools/stress_dense.py <seconds>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vsearch_amd.device_index import DeviceIndex
from oracle import compare
t_end = time.time() + float(sys.argv[1]); it = 0
while time.time() < t_end:
    rng = np.random.default_rng(1000 + it)
    n = int(rng.choice([1, 5, 130, 1000, 5000, 20000])); V = int(rng.choice([1, 31, 32, 33, 500, 999, 4097, 29523]))
    if n * V > 3e8: n = max(1, int(3e8 // V))
    B = int(rng.choice([1, 3, 32, 33, 130])); k = int(min(n, rng.choice([1, 10, 100, 700, 2500])))
    dens = float(rng.choice([1.0, 0.3, 0.02]))
    g = torch.Generator(device="cuda").manual_seed(it)
    mat = torch.randn((n, V), device="cuda", generator=g) * (torch.rand((n, V), device="cuda", generator=g) < dens)
    q = torch.randn((B, V), device="cuda", generator=g) * (torch.rand((B, V), device="cuda", generator=g) < 0.5)
    want = (q.double() @ mat.double().t()).float().cpu().numpy()
    for md in (0.0, 0.05):
        idx = DeviceIndex.from_dense(mat, max_density=md)
        ids, sc = idx.search(q, k)
        ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
        scale = float((q.double().abs() @ mat.double().abs().t()).max()) + 1e-30      # signed terms cancel: errors scale with sum |q p|
        try:
            # absolute tolerance relative to the row's score scale (signed values cancel)
            got_true = np.take_along_axis(want, ids, axis=1)
            assert np.all(np.abs(got_true - sc) <= 2e-5 * scale + 1e-6), "scores"
            assert all(len(set(r.tolist())) == k for r in ids), "dup ids"
            kth = np.sort(want, axis=1)[:, ::-1][:, k - 1]
            assert np.all(sc.min(axis=1) >= kth - 2e-5 * scale - 1e-6), "not top-k"
            assert np.all(np.diff(sc, axis=1) <= 1e-6 * scale + 1e-7), "order"
        except AssertionError as e:
            print(f"FAIL it={it} n={n} V={V} B={B} k={k} dens={dens} md={md}: {e}"); sys.exit(1)
        idx.close()
    it += 1
print("dense stress ok", it)
