#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json (`configs[1..4]` are parity cases; bench.py measures the metric's
own workload).  One JSON line per config, each with a `roofline` object, on ONE MI355X:

  c3   1 M synthetic docs sparse CSR fp32, batch 1024, k = 100          (HBM roofline)
  c5   Wiki21M-shaped binary bag-of-token index, batch 1024, k = 100    (HBM roofline; bit-exact check)
  c2   100 k docs dense fp32 [N, 29523], batch 256, k = 100             (fp32 MFMA roofline)
  ref  the reference's own torch calls (oracle/torch_ref.py) run on the same GPU through PyTorch-ROCm,
       at c3 / c2 shape where torch supports them -- "what the reference does on this chip today"

    python tools/bench_configs.py [c3 c5 c2 ref]
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402  (synthetic query generator + checker only)
from oracle import compare  # noqa: E402
from vsearch_amd import _native as nat  # noqa: E402
from vsearch_amd.device_index import DeviceIndex, Profile  # noqa: E402

V, K = 29523, 100
HBM_PEAK, MFMA_F32_PEAK = 8000.0, 157.3


def timed(fn, warmup=1, steps=3):
    for _ in range(warmup):
        fn()
    Profile.enable(True)
    Profile.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    Profile.enable(False)
    return dt, out


def sparse_like(name, n, nnz, kind, store, batch, q_law):
    idx = DeviceIndex.synthetic(0, 0, n, V, nnz, kind, 0, store)
    info = idx.info()
    q = torch.from_numpy(oracle.synth_queries(1, batch, V, 776, q_law)).cuda()
    dt, (ids, sc) = timed(lambda: idx.search(q, K))
    scan_ms, launches = Profile.read("csr_scan_topk")
    info = idx.info()
    qt = max(1, info.queries_per_pass)
    passes = -(-batch // qt)
    achieved = info.last_scan_bytes / (scan_ms / 1e3 / launches) / 1e9      # bytes the scan kernel had to read for the batch
    # exactness / validity on a sample of queries against the independent scores-only kernel
    sample = q[:4].cpu().numpy()
    allsc = idx.scores(sample)
    compare.check_topk_valid(allsc, ids[:4].cpu().numpy(), sc[:4].cpu().numpy(), rtol=1e-4, exact=(kind == 1), canonical=(kind == 1))
    return {"config": name, "metric": "queries/sec", "value": batch / dt, "ms_per_batch": dt * 1e3, "docs": n, "batch": batch, "k": K,
            "queries_per_pass": qt, "lanes_per_row": info.lanes_per_row, "index_bytes": info.device_bytes,
            # (postings paths: the walk's bytes come from L2 / Infinity Cache and what binds is the LDS scatter-add rate -- an "hbm" fraction
            #  of algorithmic bytes would exceed 1 and mean nothing; HBM traffic needs a PMC pass: tools/pmc_walk.sh)
            "roofline": {"bound": "hbm" if info.last_path < 2 else "on-chip: LDS scatter-adds (ds_add_u32, bank conflicts) + L2->L1 record loads",
                         "achieved": achieved if info.last_path < 2 else (info.last_walk_postings / (scan_ms / 1e3 / launches) / 1e9),
                         "peak": HBM_PEAK if info.last_path < 2 else 5000.0, "unit": "GB/s" if info.last_path < 2 else "Gadd/s (ds_add_u32 at random addresses, tools/microbench/lds_scatter.hip)",
                         "frac": (achieved / HBM_PEAK) if info.last_path < 2 else (info.last_walk_postings / (scan_ms / 1e3 / launches) / 5.0e12),
                         "algorithmic_GBps": achieved,
                         "kernel": "bp_walk_topk" if info.last_path >= 2 else "csr_scan_topk_mq", "avg_launch_ms": scan_ms / launches,
                         "scan_path": info.last_path, "fallback_queries": info.last_fallbacks, "postings_copy_bytes": info.aux_bytes,
                         "walk_adds_per_s": (info.last_walk_postings / (scan_ms / 1e3 / launches)) if info.last_path >= 2 else None,
                         "note": "achieved = algorithmic bytes of the scan kernel / its time (served largely by L2 / Infinity Cache: not HBM utilisation)",
                         "scan_bytes": info.last_scan_bytes, "csr_bytes_per_pass": info.bytes_per_pass, "csr_passes": passes},
            "check": "top-k valid vs csr_scan_scores on 4 queries" + (" (bit-exact, canonical ids)" if kind == 1 else " (1e-4)")}


def dense(n=100_000, batch=256):
    g = torch.Generator(device="cuda").manual_seed(0)
    mat = torch.zeros((n, V), device="cuda")
    for s in range(0, n, 10000):
        c = torch.rand((min(10000, n - s), V), device="cuda", generator=g).topk(768, dim=1).indices
        mat[s:s + c.shape[0]].scatter_(1, c, 0.01 + 3 * torch.rand(c.shape, device="cuda", generator=g))
    q = torch.zeros((batch, V), device="cuda")
    qc = torch.rand((batch, V), device="cuda", generator=g).topk(776, dim=1).indices
    q.scatter_(1, qc, 0.01 + 3 * torch.rand(qc.shape, device="cuda", generator=g))
    auto = DeviceIndex.from_dense(mat, max_density=0.05)          # sparsity-aware route (what the facade's Index uses)
    dt_auto, (ids_a, sc_a) = timed(lambda: auto.search(q, K))
    auto_info = auto.info()
    auto.close()
    idx = DeviceIndex.from_dense(mat)
    dt, (ids, sc) = timed(lambda: idx.search(q, K))
    gemm_ms, launches = Profile.read("dense_scores")
    flops = 2.0 * batch * V * n
    achieved = flops / (gemm_ms / 1e3 / launches) / 1e12
    t0 = time.perf_counter()
    ref = (q @ mat.t()).topk(K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = (q @ mat.t()).topk(K)
    torch.cuda.synchronize()
    ref_dt = time.perf_counter() - t0
    rel = ((ref.values - sc).abs() / ref.values).max().item()
    same = (ref.indices == ids).float().mean().item()
    return {"config": "c2 dense 100k x 29523 fp32", "metric": "queries/sec", "value": batch / dt, "ms_per_batch": dt * 1e3, "docs": n, "batch": batch,
            "k": K, "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK, "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK,
                                 "kernel": "dense_scores_kernel", "avg_launch_ms": gemm_ms / launches, "flops_per_launch": flops},
            "reference_on_this_gpu": {"what": "torch.matmul(q, P.t()).topk(k) through PyTorch-ROCm (index.py:91-92)", "ms_per_batch": ref_dt * 1e3,
                                      "qps": batch / ref_dt},
            "sparsity_aware_route": {"what": "same matrix stored as CSR packets (density %.3f), searched by the CSR scan" % (auto_info.nnz / (n * V)),
                                     "ms_per_batch": dt_auto * 1e3, "qps": batch / dt_auto, "queries_per_pass": auto_info.queries_per_pass,
                                     "ids_equal_to_torch_frac": (ref.indices == ids_a).float().mean().item(),
                                     "max_rel_score_err": ((ref.values - sc_a).abs() / ref.values).max().item()},
            "check": {"ids_equal_to_torch_frac": same, "max_rel_score_err": rel}}


def reference_gpu_sparse(n=1_000_000, batch=256):
    """index.py:88-94 as the reference would run it on this GPU: torch sparse-CSR matmul + topk via PyTorch-ROCm."""
    from oracle import torch_ref
    idx = DeviceIndex.synthetic(0, 0, n, V, 768, 0, 0, nat.VS_F32)
    ip, ix, d = idx.export_csr()
    idx.close()
    out = {"config": f"reference torch path on this GPU, sparse {n} docs", "batch": batch}
    try:
        vec = torch_ref.make_csr(ip, ix, d, (n, V)).cuda()
        q = torch.from_numpy(oracle.synth_queries(1, batch, V, 776)).cuda()
        torch_ref.search(vec, q, K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ids, sc = torch_ref.search(vec, q, K)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out.update({"ms_per_batch": dt * 1e3, "qps": batch / dt, "ok": True})
    except Exception as e:                                              # torch may not implement dense x CSR^T on ROCm
        out.update({"ok": False, "error": f"{type(e).__name__}: {str(e)[:300]}"})
    return out


def main():
    which = sys.argv[1:] or ["c3", "c5", "c2", "ref"]
    nat.require_device()
    for w in which:
        if w == "c3":
            r = sparse_like("c3 sparse 1M x 768 fp32", 1_000_000, 768, 0, nat.VS_F32, 1024, 0)
        elif w == "c5":
            r = sparse_like("c5 bag-of-token 21015324 docs binary", 21_015_324, 86, 1, nat.VS_NONE, 1024, 1)
        elif w == "c2":
            r = dense()
        elif w == "ref":
            r = reference_gpu_sparse()
        else:
            continue
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
