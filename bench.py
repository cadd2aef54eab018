#!/usr/bin/env python3
"""bench.py -- queries/sec of the vocabulary-space retrieval hot path on MI355X.

Metric (BASELINE.json): queries/sec over a 21 M-doc sparse CSR index (V = 29 523, 768 nnz/doc, fp32),
k = 100.  One "step" = one batch of B = 1024 synthetic queries (768 + 8 nnz each) searched against
the whole index: scoring scan + fused top-k + merge (+ one all-gather and a final merge when the
index is row-sharded over N GPUs).  The scan is the blocked-postings kernel when HBM has room for the
column-grouped copy of the shard (default; it is built as part of the index build), else the 8-query CSR
scan (`--scan csr` forces it).  Index and queries are resident in HBM when the timed region
starts.  Strong scaling: the 21 015 324-row index is re-partitioned over the N ranks.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the scan, HBM roofline) and
`cpu_baseline` (the reference's three torch calls, oracle/torch_ref.py, timed on this host).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_DOCS = 21_015_324          # Wiki21M (test/svdr_wiki21m/build_binary_token_index.sh:14)
V = 29_523
NNZ_DOC = 768
NNZ_Q = 776
BATCH = 1024
K = 100
INDEX_SEED, QUERY_SEED = 0, 1
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LDS_ADD_U32_PEAK = 7.8e12    # ds_add_u32, <= 2 lanes per bank (what the arranged quad chunks are built for): 12.83 adds / clk / CU x 256 CUs x 2.38 GHz
                             # (profiles/r04_conflicts.txt, tools/microbench/lds_conflicts.hip)
LDS_ADD_U32_RANDOM = 5.0e12  # the same at random addresses (profiles/r02_lds_scatter.txt): what the list walks of records see
L2_PEAK_GBS = 34500.0        # aggregate L2 -> L1 bandwidth, 256 CUs x 64 B / clk (MI355X_MICROARCH.md): the bound of a walk whose bytes are L2-served
SHADER_CLOCK_HZ = 2.4e9      # (cycles-per-chunk figures are quoted at this clock)
MFMA_F32_PEAK_TF = 157.3     # fp32-input MFMA = the fp32 vector peak (MI355X_MICROARCH.md)
WALK_KERNEL = {-1: "csr_scan_topk_mq", 0: "bp_walk_topk", 4: "bp_quad_topk", 5: "bp_bin_topk", 6: "bp_bq_topk"}     # vs_index_info_t.postings_walk -> the filter's kernel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--docs", type=int, default=N_DOCS, help="total index rows (default: the metric's 21 015 324)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-ids", default=None, help="rank 0 writes the last step's (ids, scores) to this .npz (tests)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (C3, C5, Zipf 21 M, small-batch latency, facade leg) "
                                                                "that follow the headline measurement on one GPU")
    ap.add_argument("--scan", choices=["auto", "csr", "postings"], default="auto",
                    help="auto: blocked postings when HBM has room for the second copy (default); csr: the 8-query CSR scan only")
    ap.add_argument("--cpu-sample-docs", type=int, default=100_000)
    ap.add_argument("--columns", choices=["uniform", "zipf"], default="uniform",
                    help="column law of the synthetic corpus AND queries: uniform (the metric's workload) or zipf (popularity ~ 1 / rank, "
                         "SURVEY 8(d) C3's secondary run: vsearch_amd/synth.py KIND_SKEW)")
    ap.add_argument("--store", choices=["fp32", "fp16"], default="fp32",
                    help="value dtype streamed by the scan (fp16 = the reference's fp16=True load default, index.py:135)")
    return ap.parse_args()


KIND = {"uniform": 0, "zipf": 2}


def kernel_source_hash():
    """sha256 over the library's sources: a PMC profile under profiles/ is only quoted for the build it was taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(REPO, "vsearch_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(src, "*.hip"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return hexd(h)


def hexd(h):
    return h.hexdigest()[:16]


def roof(bound, achieved, peak, unit, **extra):
    """a `roofline` object of a secondary leg: achieved / peak of the resource that bounds the dominant kernel of the leg"""
    return dict({"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak}, **extra)


def event_ms(fn, reps, warmup=2):
    """average GPU time of fn() in ms (events on torch's current stream -- the stream the library launches on)"""
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def make_query_batches(n_batches, batch, device, kind=0):
    """Distinct synthetic query batches, generated on the GPU by the library's own generator (rows of the
    seed-1 synthetic matrix with 776 non-zeros) and left resident in HBM as dense [B, V] fp32."""
    from vsearch_amd.device_index import DeviceIndex
    out = []
    for i in range(n_batches):
        gen = DeviceIndex.synthetic(QUERY_SEED, i * batch, batch, V, NNZ_Q, kind, 0, 0, device.index or 0)
        ip, ix, d = gen.export_csr()
        gen.close()
        q = torch.zeros((batch, V), dtype=torch.float32, device=device)
        rows = torch.from_numpy(np.repeat(np.arange(batch), np.diff(ip))).to(device)
        q[rows, torch.from_numpy(ix).to(device)] = torch.from_numpy(d).to(device)
        out.append(q)
    return out


def parity_check(device, kind=0, nnz_doc=NNZ_DOC, store=0, val_law=0, expect_path=None, n=20_000, exact=False):
    """Small prefix of the same synthetic index (rows are a pure function of (seed, row id)) searched
    by the HIP path and by the CPU oracle: recall@100 and max relative score error.  `expect_path`: the scan path the full-size
    run took -- the prefix must take the same one (3 = postings filter + refine), or the flag says nothing about it."""
    import oracle
    from oracle import compare
    from vsearch_amd import _native as nat
    from vsearch_amd.device_index import DeviceIndex
    idx = DeviceIndex.synthetic(INDEX_SEED, 0, n, V, nnz_doc, kind, 0, store, device)
    q = oracle.synth_queries(QUERY_SEED, 8, V, NNZ_Q, val_law, kind=0 if kind == 1 else kind)
    fp16 = store == nat.VS_F16
    if fp16:
        # the reference casts the query to the index dtype (index.py:89) and returns scores in it: the oracle sees the same fp16 images
        # (export_csr returns the stored fp16 values), and the comparison allows the result's own fp16 rounding (2^-11)
        q = q.astype(np.float16).astype(np.float32)
    rtol = 1e-3 if fp16 else 1e-4
    ids, sc = idx.search(q, K)
    path = idx.info().last_path
    if expect_path is not None and path != expect_path:
        raise AssertionError(f"parity prefix took scan path {path}, the measured run took {expect_path}")
    ip, ix, d = idx.export_csr()
    binary = store == nat.VS_NONE
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix.astype(np.int32), None if binary else d, V, q, K, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=rtol, exact=exact, canonical=exact)
    rel = float(np.max(np.abs(sc.astype(np.float64) - o_sc) / np.abs(o_sc)))
    idx.close()
    return {"docs": n, "queries": 8, "scan_path": path, "recall_at_100_vs_oracle": compare.recall_at_k(o_ids, ids), "max_rel_score_err": rel, "rtol": rtol,
            "ids_bit_exact": bool((np.asarray(ids) == o_ids).all()) if exact else None}


def parity_sharded(world, rank, local_rank, device, kind=0, n=None, expect_path=None):
    """N > 1: the same check THROUGH the sharded path -- every rank holds its row range of a small synthetic index, the query batch
    goes through ShardedSearcher.search (local search, the one all-gather, merge), and rank 0 compares the merged result with the
    CPU oracle's over the whole index (rows regenerated on the host: a pure function of (seed, row id))."""
    import oracle
    from oracle import compare
    from vsearch_amd.device_index import DeviceIndex
    from vsearch_amd.distributed import ShardedSearcher, shard_rows
    n = n or 20_000 * world              # every rank's shard is big enough for the path the measured run takes (the postings filter)
    row0, n_loc = shard_rows(n, world, rank)
    idx = DeviceIndex.synthetic(INDEX_SEED, row0, n_loc, V, NNZ_DOC, kind, 0, 0, local_rank)
    searcher = ShardedSearcher.from_device_index(idx, row0, n)
    q = oracle.synth_queries(QUERY_SEED, 8, V, NNZ_Q, 0, kind=kind)
    ids, sc = searcher.search(torch.from_numpy(q).to(device), K)
    ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
    path = idx.info().last_path
    idx.close()
    if expect_path is not None and path != expect_path:
        raise AssertionError(f"rank {rank}: the sharded parity index took scan path {path}, the measured run took {expect_path}")
    if rank != 0:
        return None
    ip, ix, d = oracle.synth_csr(INDEX_SEED, 0, n, V, NNZ_DOC, kind)
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix.astype(np.int32), d, V, q, K, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=1e-4)
    rel = float(np.max(np.abs(sc.astype(np.float64) - o_sc) / np.abs(o_sc)))
    return {"docs": n, "docs_per_rank": n_loc, "queries": 8, "ranks": world, "scan_path_rank0": path, "through": "ShardedSearcher.search (all-gather + merge)",
            "recall_at_100_vs_oracle": compare.recall_at_k(o_ids, ids), "max_rel_score_err": rel}


def cpu_baseline(sample_docs, device, kind=0):
    """The reference's Index.search (index.py:89-92: cast, torch.matmul(q, csr.t()), topk) restated in
    oracle/torch_ref.py and timed on this box's host cores on a bounded sample of the same index."""
    from oracle import torch_ref
    from vsearch_amd.device_index import DeviceIndex
    n = sample_docs
    idx = DeviceIndex.synthetic(INDEX_SEED, 0, n, V, NNZ_DOC, kind, 0, 0, device)   # bit-identical rows, generated on the GPU
    ip, ix, d = idx.export_csr()
    idx.close()
    vec = torch_ref.make_csr(ip, ix, d, (n, V))
    import oracle
    bq = 32
    q = torch.from_numpy(oracle.synth_queries(QUERY_SEED, bq, V, NNZ_Q, kind=kind))
    torch_ref.search(vec, q, K)                                   # warm-up
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t0 < 10.0 and reps < 50):
        torch_ref.search(vec, q, K)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    qps_sample = bq / dt
    return {"value": qps_sample * n / N_DOCS, "unit": "queries/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"reference Index.search restated (torch {torch.__version__} CPU sparse-CSR matmul + topk, fp32): "
                      f"{n} of the same synthetic docs x 768 nnz, {bq}-query batches, {reps} timed calls, "
                      f"{qps_sample:.1f} q/s on the sample, scaled linearly in nnz to {N_DOCS} docs",
            "sample_qps": qps_sample, "host_cpus": os.cpu_count()}


# (VS_BENCH_LEGS name, key in `secondary`, (docs, nnz per doc, synth kind, store dtype, query value law, steps, keyword arguments))
SECONDARY_LEGS = (
    ("C3", "C3_1m_sparse", (1_000_000, NNZ_DOC, 0, 0, 0, 10, {})),
    ("C5", "C5_bot_21m", (N_DOCS, 86, 1, -1, 1, 5, {"exact": True})),                        # SVDR bag-of-token index, dyadic query weights: bit-exact
    ("zipf", "zipf_21m", (N_DOCS, NNZ_DOC, 2, 0, 0, 3, {"columns": "zipf"})),
    ("fp16", "fp16_21m", (N_DOCS, NNZ_DOC, 0, 1, 0, 3, {})),                                 # the reference's load default (index.py:135 fp16=True)
)


def summarise(line):
    """{leg: [ms per step, queries/sec (or null), roofline frac, parity ok]} of the headline and every secondary leg, <= 1 500 characters,
    the LAST key of the JSON line (the driver keeps known keys + the line's tail)."""
    def ok(p):
        if not p:
            return None
        if "ok" in p:
            return bool(p["ok"])
        if p.get("ids_bit_exact") is not None:
            return bool(p["ids_bit_exact"])
        return bool(p.get("recall_at_100_vs_oracle", 0) >= 0.999 and p.get("max_rel_score_err", 1) <= p.get("rtol", 1e-4))
    r3 = lambda x: None if x is None else float(f"{x:.4g}")
    out = {"_": "leg: [ms, q/s, roofline.frac, parity ok]", "headline": [r3(line["ms_per_step"]), r3(line["value"]), r3(line["roofline"]["frac"]), ok(line.get("parity"))]}
    short = {"C2_dense_100k": "C2", "C3_1m_sparse": "C3", "C5_bot_21m": "C5", "zipf_21m": "zipf", "fp16_21m": "fp16", "facade": "facade", "shard_group_8_on_one_gpu": "shards8",
             "deep_k_21m": "deepk", "embed_mask_B1024": "mask", "dense_to_csr_B1024": "to_csr", "embed_to_csr_B1024": "mask_csr", "head_project_pool_64x256": "head", "rerank_1024x100": "rerank"}
    for key, rec in (line.get("secondary") or {}).items():
        if key not in short or not isinstance(rec, dict) or "error" in rec:
            continue
        ms = rec.get("ms_per_step", rec.get("kernel_ms", rec.get("ms")))
        out[short[key]] = [r3(ms), r3(rec.get("queries_per_sec")), r3((rec.get("roofline") or {}).get("frac")), ok(rec.get("parity"))]
    if (line.get("secondary") or {}).get("errors"):
        out["errors"] = len(line["secondary"]["errors"])
    return out


def _timed(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def pmc_leg(name, batch):
    """HBM bytes per launch of a secondary leg's scan kernel from the committed rocprofv3 --pmc passes (profiles/pmc_summary.json ->
    "legs"), only when taken on this kernel build and batch size; else None."""
    try:
        rec = json.load(open(os.path.join(REPO, "profiles", "pmc_summary.json"))).get("legs", {}).get(name)
        if rec and rec.get("source_hash") == kernel_source_hash() and rec.get("queries_per_launch") == batch and rec.get("hbm_bytes_per_launch"):
            return rec
    except Exception:
        pass
    return None


def pmc_utilisation(kernel):
    """What the walk waits for, from the committed counter passes (tools/pmc_walk.sh -> tools/pmc_walk_summary.py -> profiles/pmc_summary.json
    "utilisation"), only when taken on this kernel build: L1 -> L2 read requests per clock and CU, the L2 hit rate, and the request rate as a
    fraction of the ceiling tools/microbench/l2_requests.hip measured for that hit rate (0.40 requests a clock and CU from L2, 0.095 beyond)."""
    try:
        rec = json.load(open(os.path.join(REPO, "profiles", "pmc_summary.json"))).get("utilisation", {}).get(kernel)
        if rec and rec.get("source_hash") == kernel_source_hash():
            return {k: rec[k] for k in ("l2_requests_per_clk_cu", "l2_hit_rate", "request_ceiling_frac", "tcp_tcc_read_latency_cycles", "shape", "tag") if k in rec}
    except Exception:
        pass
    return None


def run_leg(name, docs, nnz, kind, store, val_law, steps, B, k, local_rank, device, exact=False, columns="uniform", parity=True):
    """One secondary configuration: builds the synthetic index, times `steps` searches of 1024-query batches, returns the record
    (ms, q/s, scan kernel time, `roofline`, oracle parity of a prefix).  The `roofline` is traffic-based when profiles/pmc_summary.json
    holds FETCH_SIZE / WRITE_SIZE passes of this leg on this kernel build; otherwise the kernel's algorithmic bytes are priced
    against the L2 -> L1 aggregate (they are L2 / Infinity-Cache served: never against the HBM peak)."""
    from vsearch_amd import _native as nat
    from vsearch_amd import synth
    from vsearch_amd.device_index import DeviceIndex, Profile
    t0 = time.perf_counter()
    idx = DeviceIndex.synthetic(INDEX_SEED, 0, docs, V, nnz, kind, 0, store, local_rank)
    qkind = 0 if kind == synth.KIND_BOT else kind
    qs = []
    for i in range(2):
        gen = DeviceIndex.synthetic(QUERY_SEED, i * B, B, V, NNZ_Q, qkind, val_law, 0, local_rank)
        ip, ix, d = gen.export_csr()
        gen.close()
        q = torch.zeros((B, V), dtype=torch.float32, device=device)
        q[torch.from_numpy(np.repeat(np.arange(B), np.diff(ip))).to(device), torch.from_numpy(ix).to(device)] = torch.from_numpy(d).to(device)
        qs.append(q)
    idx.search(qs[0][:8], k)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    it = [0]

    def step():
        idx.search(qs[it[0] % 2], k)
        it[0] += 1
    Profile.enable(True)
    Profile.reset()
    dt = _timed(step, steps)
    Profile.enable(False)
    scan_ms, launches = Profile.read("csr_scan_topk")
    pre_ms, _ = Profile.read("head_gemm")
    info = idx.info()
    launch_s = scan_ms / 1e3 / max(1, launches)                      # the scan launches of one search (a head-column corpus: one per half batch)
    per_search_s = scan_ms / 1e3 / max(1, steps + 1)
    algo = info.last_scan_bytes / per_search_s / 1e9 if per_search_s > 0 else 0.0
    kernel = WALK_KERNEL.get(info.postings_walk, "?")
    pmc = pmc_leg(name, B)
    one_pass = info.bytes_per_pass / (HBM_PEAK_GBS * 1e9)
    extra = dict(one_pass_lower_bound_ms=one_pass * 1e3, frac_of_one_pass_lower_bound=one_pass / per_search_s if per_search_s > 0 else None,
                 algorithmic_GBps=algo, kernel=kernel)
    if info.last_path >= 2 and info.head_columns == 0 and per_search_s > 0:
        # (with head columns the strips' multiply-adds run on the matrix cores: not LDS adds)
        peak = LDS_ADD_U32_PEAK if info.postings_walk == 4 else LDS_ADD_U32_RANDOM
        extra["frac_of_lds_add_peak"] = info.last_walk_postings / per_search_s / peak
        extra["lds_add_peak_per_s"] = peak
    if pmc:
        # (a corpus with head columns: the profiled bytes are the list walk's AND the head pre-pass product's -- so is the time)
        t_pmc = per_search_s + (pre_ms / 1e3 / max(1, steps + 1) if any("head_gemm" in k for k in pmc.get("kernels", [])) else 0.0)
        hbm = pmc["hbm_bytes_per_launch"] / t_pmc / 1e9
        rl = roof("hbm", hbm, HBM_PEAK_GBS, "GB/s", traffic=pmc["hbm_bytes_per_launch"],
                  achieved_is="HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this leg, profiles/pmc_summary.json) / scan kernel time (+ the head pre-pass product's, where there is one)",
                  traffic_source=pmc.get("source"), **extra)
    else:
        rl = roof("hbm", algo, L2_PEAK_GBS, "GB/s", traffic=None, bound_note="L2: the walk's bytes are L2 / Infinity-Cache served; no PMC pass of this leg on this kernel "
                  "build, so the algorithmic bytes are priced against the L2 -> L1 aggregate, not the HBM peak",
                  achieved_is="algorithmic bytes of the scan kernel / its time", **extra)
    ut = pmc_utilisation("bp_bq_topk" if info.postings_walk == 6 else "bp_quad_topk" if info.postings_walk == 4 else "bp_walk_topk") if info.last_path == 3 else None
    if ut:
        for key in ("l2_requests_per_clk_cu", "l2_hit_rate", "request_ceiling_frac"):
            rl[key] = ut.get(key)
        rl["utilisation_source"] = f"profiles/pmc_summary.json utilisation ({ut.get('tag')}, {ut.get('shape')})"
    rec = {"docs": docs, "nnz_per_doc": nnz, "columns": columns, "store": {0: "fp32", 1: "fp16", -1: "binary"}.get(store, str(store)), "batch": B, "k": k, "steps": steps,
           "ms_per_step": dt * 1e3, "queries_per_sec": B / dt,
           "scan_path": info.last_path, "kernel": kernel, "scan_kernel_ms": per_search_s * 1e3, "scan_launches_per_search": launches / max(1, steps + 1),
           "head_gemm_ms": pre_ms / max(1, steps + 1) if pre_ms else None,
           "fallback_queries": info.last_fallbacks, "head_columns": info.head_columns, "postings_copy_bytes": info.aux_bytes,
           "walk_adds_per_s": (info.last_walk_postings / per_search_s) if per_search_s > 0 and info.last_path >= 2 else None,
           "roofline": rl, "index_build_s": round(build_s, 2)}
    idx.close()
    if parity:
        rec["parity"] = parity_check(local_rank, kind, nnz, store, val_law, expect_path=info.last_path, n=80_000 if kind == synth.KIND_BOT else 20_000, exact=exact)
    return rec


def secondary(index, batches, args, local_rank, device, headline_s):
    """The other configurations of BASELINE.json and the drop-in API, AFTER the headline's timed region (never inside it), each
    with its own oracle-prefix parity flag: so that they stand under the driver's clock too.
      facade      -- the same 21 M-doc index searched through the reference's API (SparseIndex.search: cast, device move, SearchResults)
      latency     -- B = 1 and B = 32 on the same index
      C3          -- 1 M docs x 768 nnz, B = 1024
      C5          -- SVDR bag-of-token index, 21 M docs x ~86 binary nnz, dyadic query weights (bit-exact ids)
      zipf_21m    -- the 21 M-doc index with Zipf column popularity (dense head strips on the matrix cores)"""
    from vsearch_amd import _native as nat
    from vsearch_amd import synth
    from vsearch_amd.device_index import DeviceIndex, Profile
    from vsearch_amd.ir.retriever.index import SparseIndex
    out = {}
    B = args.batch
    only = [x for x in os.environ.get("VS_BENCH_LEGS", "").split(",") if x]      # developer: run only these legs (facade,shards,C3,C5,zipf)
    want = lambda name: not only or name in only
    # facade leg: Index.search of the reference's API (index.py:88-94) on the SAME device index
    if want("facade"):
        fac = SparseIndex()
        fac.adopt_device_index(index, dtype=torch.float32)
        t_direct = _timed(lambda: index.search(batches[0], args.k), 3)
        t_facade = _timed(lambda: fac.search(batches[0], args.k), 3)
        fac._dev = None                                               # (the bench owns the index)
        out["facade"] = {"api": "vsearch_amd.ir.retriever.index.SparseIndex.search (reference: src/ir/retriever/index.py:88-94)", "ms_per_step": t_facade * 1e3,
                         "device_index_ms_per_step": t_direct * 1e3, "overhead_frac": t_facade / t_direct - 1.0, "queries_per_sec": B / t_facade}
        lat = {}
        for b in (1, 32):
            lat[f"B={b}_ms"] = _timed(lambda: index.search(batches[0][:b], args.k), 20, 3) * 1e3
        out["latency_21m"] = lat
    # deep k (reference: index.py:92 takes any k <= N): beyond the filter's candidate buffers (k + margin > 1024 ranks) a search on the quad
    # copy leaves it for "search after" passes of the CSR scan; exact records (postings_walk = 0) serve 1024 ranks a pass on the fp64 walk
    # (DESIGN 2).  64 queries, k = 2000, the same 21 M-doc index -- the documented cliff, with a number (ADVICE r4 / VERDICT r5 item 7)
    if want("deepk"):
        try:
            kd, bd = 2000, 64
            qd = batches[0][:bd]
            ids_a, sc_a = index.search(qd, kd)
            t_quad = _timed(lambda: index.search(qd, kd), 2)
            path_a = index.info().last_path
            index.set_option("postings_walk", 0)                      # records instead of quad chunks (rebuilt at the next search) ...
            index.set_option("postings_quant", 0)                     # ... with exact fp32 values: what the fp64 walk needs
            ids_b, sc_b = index.search(qd, kd)
            t_rec = _timed(lambda: index.search(qd, kd), 2)
            path_b = index.info().last_path
            same = bool((ids_a == ids_b).all().item() and (sc_a == sc_b).all().item())
            out["deep_k_21m"] = {"k": kd, "batch": bd, "docs": N_DOCS, "ms_per_step": t_quad * 1e3, "queries_per_sec": bd / t_quad,
                                 "quad_copy": {"ms_per_step": t_quad * 1e3, "queries_per_sec": bd / t_quad, "last_path": int(path_a),
                                               "note": "k + margin beyond the filter's 1024-rank buffers: CSR scan, 'search after' passes"},
                                 "record_copy": {"ms_per_step": t_rec * 1e3, "queries_per_sec": bd / t_rec, "last_path": int(path_b),
                                                 "note": "postings_walk = 0, postings_quant = 0: exact fp32 records, fp64 walk, 1024 ranks a pass"},
                                 "k100_same_batch_ms": _timed(lambda: index.search(qd, args.k), 2) * 1e3,
                                 "parity": {"vs": "the two paths against each other (ids and scores bit for bit)", "ok": same}}
            index.set_option("postings_walk", -1)
            index.set_option("postings_quant", -1)
        except Exception as e:
            out["deep_k_21m"] = {"error": str(e)[:200]}
    index.close()

    # one process, 8 row shards of the same index on this one device (vs_shard_group_*: per-shard searches on their own streams, the
    # B * k pairs gathered and merged on the first shard's device): what the in-process sharding adds to 8 x the single-shard step
    try:
        if not want("shards"):
            raise KeyError("skipped")
        from vsearch_amd.device_index import ShardGroup
        from vsearch_amd.distributed import shard_rows
        t0 = time.perf_counter()
        shards = []
        for r in range(8):
            row0, n_loc = shard_rows(N_DOCS, 8, r)
            shards.append(DeviceIndex.synthetic(INDEX_SEED, row0, n_loc, V, NNZ_DOC, 0, 0, nat.VS_F32, local_rank))
        grp = ShardGroup(shards)
        grp.search(batches[0][:8], args.k)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        t_one = _timed(lambda: shards[3].search(batches[0], args.k), 3)
        t_grp = _timed(lambda: grp.search(batches[0], args.k), 2)
        grp.close()
        # the same 8 row shards through the reference's API: SparseIndex(..., devices=[...]) keeps a shard group behind Index.search
        fs = SparseIndex()
        fs._dtype, fs._shape = torch.float32, (N_DOCS, V)
        fs._adopt_shards(shards)
        t_fac = _timed(lambda: fs.search(batches[0], args.k), 2)
        out["shard_group_8_on_one_gpu"] = {"shards": 8, "docs_per_shard": shard_rows(N_DOCS, 8, 0)[1], "ms_per_step": t_grp * 1e3, "single_shard_ms_per_step": t_one * 1e3,
                                           "overhead_ms_per_step": (t_grp - 8 * t_one) * 1e3, "queries_per_sec": B / t_grp, "index_build_s": round(build_s, 2),
                                           "note": "all 8 shards share ONE GPU here: the searches serialise; on 8 GPUs a step is the slowest shard's step + the exchange"}
        out["facade_sharded"] = {"api": "vsearch_amd.ir.retriever.index.SparseIndex(index_file='shard*.npz', devices=[...]).search -> vs_shard_group_search "
                                        "(reference: src/ir/retriever/index.py:88-94 on the vstack of its shards, :172-175)", "shards": 8, "devices": "8 x this one GPU",
                                 "ms_per_step": t_fac * 1e3, "shard_group_ms_per_step": t_grp * 1e3, "overhead_frac": t_fac / t_grp - 1.0, "queries_per_sec": B / t_fac}
        fs._drop_device()                                            # closes the group and the shards
    except Exception as e:                                       # (never lose the headline line to a secondary leg)
        out["shard_group_8_on_one_gpu"] = {"error": str(e)[:200]}
    for name, key, leg_args in SECONDARY_LEGS:
        if not want(name):
            continue
        try:
            out[key] = run_leg(key, *leg_args[:6], B, args.k, local_rank, device, **leg_args[6])
        except Exception as e:
            out.setdefault("errors", []).append(f"{name}: {str(e)[:300]}")
    # ---- the rest of the path under the same clock: dense index (C2), sparsify kernels, fused encoder head, rerank
    def c2_dense():
        n, b = 100_000, 256
        g = torch.Generator(device=device).manual_seed(0)
        mat = torch.zeros((n, V), device=device)
        for s0 in range(0, n, 10000):
            c = torch.rand((min(10000, n - s0), V), device=device, generator=g).topk(NNZ_DOC, dim=1).indices
            mat[s0:s0 + c.shape[0]].scatter_(1, c, 0.01 + 3 * torch.rand(c.shape, device=device, generator=g))
        q = torch.zeros((b, V), device=device)
        qc = torch.rand((b, V), device=device, generator=g).topk(NNZ_Q, dim=1).indices
        q.scatter_(1, qc, 0.01 + 3 * torch.rand(qc.shape, device=device, generator=g))
        idx = DeviceIndex.from_dense(mat)
        ids, sc = idx.search(q, args.k)
        Profile.enable(True)
        Profile.reset()
        dt = _timed(lambda: idx.search(q, args.k), 5)
        Profile.enable(False)
        gemm_ms, launches = Profile.read("dense_scores")
        flops = 2.0 * b * V * n
        # parity against the CPU oracle, on a row sample: for 8 queries, every RETURNED row and 1 500 random rows are
        # scored by oracle.dense_search (fp64 sums): the returned scores must be the oracle's (1e-4) and no sampled row may beat a query's
        # k-th returned score unless it was returned
        import oracle
        nq_chk = 8
        g_ids = ids[:nq_chk].cpu().numpy()
        g_sc = sc[:nq_chk].cpu().numpy()
        rows = np.unique(np.concatenate([g_ids.reshape(-1), np.random.default_rng(0).choice(n, 1500, replace=False)]))
        sub = mat[torch.from_numpy(rows).to(device)].cpu().numpy()
        o_ids, o_sc = oracle.dense_search(sub, q[:nq_chk].cpu().numpy(), sub.shape[0], acc64=True)
        rel, beaten = 0.0, 0
        for i in range(nq_chk):
            score_of = dict(zip(rows[o_ids[i]].tolist(), o_sc[i].tolist()))
            want = np.array([score_of[int(r)] for r in g_ids[i]], dtype=np.float64)
            rel = max(rel, float(np.max(np.abs(want - g_sc[i]) / np.abs(want))))
            kth = float(g_sc[i, -1])
            better = rows[o_ids[i][o_sc[i] > kth * (1 + 1e-4)]]
            beaten += int(np.setdiff1d(better, g_ids[i]).size)
        idx.close()
        out["C2_dense_100k"] = {"docs": n, "batch": b, "k": args.k, "ms_per_step": dt * 1e3, "queries_per_sec": b / dt, "kernel": "dense_scores_kernel (v_mfma_f32_32x32x2_f32)",
                                "roofline": roof("mfma", flops / (dt * 1e12), MFMA_F32_PEAK_TF, "TFLOP/s", achieved_is="2 B V N flop / step time (dense kernels + select; fp32 in, fp32 accumulate)"),
                                "parity": {"vs": "oracle.dense_search (CPU, fp64 sums) on the returned rows + 1 500 random rows, 8 queries", "max_rel_score_err": rel,
                                           "sampled_rows_beating_kth_not_returned": beaten, "ok": bool(rel <= 1e-4 and beaten == 0)}}

    def kernel_ms(scope, fn, reps, warmup=2):
        """(per-call GPU time incl. the host gaps between calls, kernel time of the library's `scope` per call: hipEvents around the launches on their stream)"""
        call = event_ms(fn, reps, warmup)
        Profile.enable(True)
        Profile.reset()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        Profile.enable(False)
        tot, _ = Profile.read(scope)
        return call, tot / reps

    def sparsify():
        from vsearch_amd.ir.utils import sparse as sp
        VOC, SHIFT, L = 30522, 999, 128
        g = torch.Generator(device=device).manual_seed(0)
        emb = torch.rand((B, V), device=device, generator=g) * 3
        tok = torch.randint(SHIFT, VOC, (B, L), device=device, generator=g)
        e2 = emb.clone()
        ms, kms = kernel_ms("mask_rows", lambda: sp.apply_embed_mask_(e2, tok, VOC, SHIFT, NNZ_DOC, True), 50)
        byts = 2.0 * B * V * 4
        out["embed_mask_B1024"] = {"what": "VDREncoder.embed mask stage (vdr.py:152-169): top-768 | lexical mask applied in place to [1024, 29523] fp32", "ms": ms,
                                   "kernel_ms": kms, "kernel": "mask_rows_fast_kernel",
                                   "roofline": roof("hbm", byts / (kms * 1e6), HBM_PEAK_GBS, "GB/s", achieved_is="one read + one write of [B, V] fp32 / kernel time (`ms`: per call, with "
                                                    "the flag read-back and the host's gap between calls)")}
        sparse = sp.topk_sparsify(emb, NNZ_DOC)
        ms, kms = kernel_ms("dense_to_csr", lambda: sp.dense_to_csr(sparse), 50)
        byts = B * V * 4.0 + B * NNZ_DOC * 8.0
        out["dense_to_csr_B1024"] = {"what": "Tensor.to_sparse_csr() of the sparsified batch (retriever.py:304)", "ms": ms, "kernel_ms": kms, "kernel": "count_nz / scan_counts / fill_csr",
                                     "roofline": roof("hbm", byts / (kms * 1e6), HBM_PEAK_GBS, "GB/s", achieved_is="one read of [B, V] fp32 + the CSR written / kernel time (the kernels read the matrix twice)")}
        # the two stages fused: the mask kernel ranks the kept elements and writes the CSR itself -- one read of [B, V]
        ms, kms = kernel_ms("mask_to_csr", lambda: sp.embed_mask_to_csr(emb, tok, VOC, SHIFT, NNZ_DOC, True), 50)
        byts = B * V * 4.0 + B * (NNZ_DOC + L) * 8.0 * 2
        out["embed_to_csr_B1024"] = {"what": "mask stage + to_sparse_csr() fused (vdr.py:152-169 + retriever.py:304): [1024, 29523] fp32 in, CSR out; replaces the two legs above in build_index",
                                     "ms": ms, "kernel_ms": kms, "kernel": "mask_rows_fast_kernel<G, 1> (one launch: select, rank, emit, place)",
                                     "roofline": roof("hbm", byts / (kms * 1e6), HBM_PEAK_GBS, "GB/s", achieved_is="one read of [B, V] fp32 + the slot runs written and compacted / kernel time")}
        Bh, Lh, H = 64, 256, 768
        hid = torch.randn((Bh, Lh, H), device=device, generator=g)
        w = torch.randn((V, H), device=device, generator=g) * 0.05
        ms = event_ms(lambda: sp.head_project_pool(hid, w), 5, 1)
        flops = 2.0 * Bh * Lh * H * V
        out["head_project_pool_64x256"] = {"what": "fused encoder head (vdr.py:70-75): LN(hidden) . W^T -> max over positions -> elu1p, [64, 256, 768] x [29523, 768]", "ms": ms,
                                           "kernel": "dense_scores_kernel (pool mode)", "roofline": roof("mfma", flops / (ms * 1e9), MFMA_F32_PEAK_TF, "TFLOP/s", achieved_is="2 B L H V flop / time (fp32 MFMA)")}

    def rerank():
        import ctypes as C
        from vsearch_amd.device_index import current_stream
        k, chunk = args.k, 8192
        g = torch.Generator(device=device).manual_seed(0)
        q = torch.zeros((B, V), device=device)
        qc = torch.rand((B, V), device=device, generator=g).topk(NNZ_Q, dim=1).indices
        q.scatter_(1, qc, 0.01 + 3 * torch.rand(qc.shape, device=device, generator=g))
        p_emb = torch.zeros((chunk, V), device=device)
        pc = torch.rand((chunk, V), device=device, generator=g).topk(NNZ_DOC, dim=1).indices
        p_emb.scatter_(1, pc, 0.01 + 3 * torch.rand(pc.shape, device=device, generator=g))
        scores = torch.empty((B, k), dtype=torch.float32, device=device)
        ids = torch.arange(B * k, device=device, dtype=torch.int64).reshape(B, k)
        o_ids, o_sc = torch.empty_like(ids), torch.empty_like(scores)
        dev = device.index or 0

        def run_all():
            st = current_stream(dev)
            for r0 in range(0, B * k, chunk):
                rows = min(chunk, B * k - r0)
                nat.check(nat.lib().vs_rerank_scores(C.c_void_p(p_emb.data_ptr()), nat.VS_F32, V, rows, r0, C.c_void_p(q.data_ptr()), V, B, k, V,
                                                     C.c_void_p(scores.data_ptr()), dev, st))
            nat.check(nat.lib().vs_rerank_topk(C.c_void_p(scores.data_ptr()), C.c_void_p(ids.data_ptr()), B, k, C.c_void_p(o_ids.data_ptr()),
                                               C.c_void_p(o_sc.data_ptr()), dev, st))
        ms = event_ms(run_all, 3, 1)
        byts = float(B) * k * V * 4
        out["rerank_1024x100"] = {"what": "retrieve(rerank=True) scoring stage (retriever.py:137-147): 102 400 re-embedded passages [.., 29523] fp32 against their queries + stable top-k",
                                  "ms": ms, "kernel": "rerank_scores_kernel + rerank_topk_kernel",
                                  "roofline": roof("hbm", byts / (ms * 1e6), HBM_PEAK_GBS, "GB/s", achieved_is="the dense passage rows read once / time")}

    for name, leg in (("C2", c2_dense), ("sparsify", sparsify), ("rerank", rerank)):
        if not want(name):
            continue
        try:
            leg()
        except Exception as e:
            out.setdefault("errors", []).append(f"{name}: {str(e)[:300]}")
        torch.cuda.empty_cache()
    out["headline_ms_per_step"] = headline_s * 1e3
    return out


def main():
    args = parse()
    # `python bench.py --gpus N` from a plain shell: this process has not touched the GPU yet; it becomes the parent of N
    # fresh rank processes (vsearch_amd/launch.py) and exits with their code.  Under torch.distributed.run we already are a rank.
    from vsearch_amd.launch import relaunch_as_ranks
    relaunch_as_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # VS_BENCH_SHARE_GPU=1 (tests on a 1-GPU box): every rank uses cuda:0 and the exchange runs over gloo -- RCCL refuses two
    # ranks on one device.  Never set for a measurement.
    share_gpu = os.environ.get("VS_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    import torch.distributed as dist
    from vsearch_amd import _native as nat
    from vsearch_amd.device_index import DeviceIndex, Profile
    from vsearch_amd.distributed import ShardedSearcher, shard_rows
    nat.require_device()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    backend = "gloo" if share_gpu else "nccl"
    if world > 1:
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    row0, n_local = shard_rows(args.docs, world, rank)
    t0 = time.perf_counter()
    store = nat.VS_F16 if args.store == "fp16" else nat.VS_F32
    kind = KIND[args.columns]
    index = DeviceIndex.synthetic(INDEX_SEED, row0, n_local, V, NNZ_DOC, kind, 0, store, local_rank)
    index.set_option("blocked_postings", {"auto": -1, "csr": 0, "postings": 1}[args.scan])
    batches = make_query_batches(min(4, args.steps + args.warmup), args.batch, device, kind)
    index.search(batches[0][:8], min(args.k, n_local))            # first sparse search builds the column-grouped copy: part of the build
    torch.cuda.synchronize()
    info = index.info()
    build_s = time.perf_counter() - t0
    searcher = ShardedSearcher.from_device_index(index, row0, args.docs)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        return searcher.search(batches[i % len(batches)], args.k)

    for i in range(args.warmup):
        step(i)
    Profile.enable(True)
    Profile.reset()
    searcher.enable_timing(True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ids, scores = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    Profile.enable(False)
    phases = searcher.read_timing()
    searcher.enable_timing(False)
    scan_ms, scan_launches = Profile.read("csr_scan_topk")
    merge_ms, _ = Profile.read("merge_topk")
    refine_ms, _ = Profile.read("refine_topk")
    fb_ms, _ = Profile.read("exact_fallback")
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # per-rank breakdown (ms per step): the walk and refine kernels (hipEvents inside the library), the whole local search, the
        # exchange (one all-gather) and the final merge (CUDA events around the phases of ShardedSearcher.search)
        mine = torch.tensor([scan_ms, refine_ms, phases["local_ms"], phases["exchange_ms"], phases["merge_ms"]], dtype=torch.float64) / max(1, args.steps)
        mine = mine.to("cpu" if share_gpu else device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        m = torch.stack(allr).cpu().numpy()
        names = ["walk_ms", "refine_ms", "local_search_ms", "exchange_ms", "merge_ms"]
        per_rank = {n: {"max": float(m[:, i].max()), "min": float(m[:, i].min())} for i, n in enumerate(names)}
        per_rank["by_rank"] = [{n: float(m[r, i]) for i, n in enumerate(names)} for r in range(world)]
    if rank == 0 and args.dump_ids:
        np.savez(args.dump_ids, ids=ids.cpu().numpy(), scores=scores.cpu().numpy())
    # N > 1: the oracle check through the sharded path (all ranks search, rank 0 compares); after the timed region
    sharded_parity = parity_sharded(world, rank, local_rank, device, kind, expect_path=index.info().last_path if n_local >= 20_000 and args.scan == "auto" else None) if world > 1 else None

    if rank == 0:
        qps = args.steps * args.batch / elapsed
        # dominant kernel = the scan ("csr_scan_topk" profile scope): bp_scan_topk on the blocked-postings path, csr_scan_topk_mq
        # on the CSR path.  Algorithmic bytes = what that kernel has to read for this batch (vs_index_info.last_scan_bytes):
        #   CSR path:      passes x bytes_per_pass                                   (every pass streams the whole index)
        #   postings path: the posting lists of the batch's (query, column) entries + one directory pair per entry and block
        info = index.info()
        qt = max(1, info.queries_per_pass)
        launches_per_step = max(1, scan_launches) / args.steps
        algo_bytes_per_launch = info.last_scan_bytes / launches_per_step
        avg_launch_s = scan_ms / 1e3 / max(1, scan_launches)
        achieved = algo_bytes_per_launch / avg_launch_s / 1e9
        path = {0: "csr scan, one query per pass", 1: "csr scan, 8 queries per pass", 2: "blocked postings, fp64 walk, 4 queries per tile",
                3: ("quad chunks" if info.postings_walk == 4 else "blocked postings") + ": int32 fixed-point filter walk (8 queries per tile) + exact refine of k+28 candidates"}[info.last_path]
        kernel = {0: "csr_scan_topk_wave", 1: "csr_scan_topk_mq", 2: "bp_walk_topk", 3: WALK_KERNEL.get(info.postings_walk, "bp_walk_topk")}[info.last_path]
        csr_equiv = -(-args.batch // qt) * info.bytes_per_pass / launches_per_step / avg_launch_s / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(REPO, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                # PMC counters cannot be read in-process: HBM bytes of this kernel come from the committed rocprofv3 --pmc
                # FETCH_SIZE / WRITE_SIZE passes over the same 21 M-doc index and batch size (profiles/pmc_summary.json)
                rec = json.load(open(pmc)).get(kernel, {})
                # only when the profile was taken on this kernel build and this configuration (it goes stale otherwise)
                if rec.get("hbm_bytes_per_launch") and rec.get("queries_per_launch") == args.batch and rec.get("docs") == n_local \
                        and rec.get("store", "fp32") == args.store and rec.get("scan", "auto") == args.scan and rec.get("columns", "uniform") == args.columns \
                        and rec.get("source_hash") == kernel_source_hash():
                    traffic = rec["hbm_bytes_per_launch"]
                    traffic_src = f"profiles/pmc_summary.json ({rec.get('tag', '?')}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command)"
            except Exception:
                traffic = None
        index_pass_bytes = info.bytes_per_pass                     # one pass over the shard's CSR packets (SURVEY 8(d) bytes_pass)
        hbm_frac = (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None
        adds_per_s = info.last_walk_postings / launches_per_step / avg_launch_s if info.last_path >= 2 else None
        # (the contract's vocabulary: "hbm" | "mfma".  This is byte / index work, priced against HBM; what the walk actually waits for is
        #  in `bound_note`)
        bound = "hbm"
        bound_note = None
        if info.last_path >= 2 and (hbm_frac or 0.0) <= 0.5:
            bound_note = ("on-chip: the walk's VALU / LDS-atomic / L1 issue together (each ~ 2/3 busy; DESIGN 4 and 8.1), not HBM -- the HBM fraction "
                          "is what the counters evidence, not what limits the kernel")
        # `achieved` / `frac`: the HBM rate the counters evidence when a matching PMC profile exists (the honest HBM fraction);
        # without one, the algorithmic rate (bytes the kernel has to read / time), flagged as such -- on the postings path most of
        # those bytes come from L2 / Infinity Cache, so that rate can exceed the HBM peak and says nothing about HBM utilisation.
        hbm_rate = (traffic / avg_launch_s / 1e9) if traffic else None
        # (no PMC profile of this kernel build: the algorithmic bytes are L2 / Infinity-Cache served -- priced against the L2 -> L1 aggregate,
        #  never against the HBM peak)
        l2_basis = hbm_rate is None and info.last_path >= 2
        if l2_basis:
            bound_note = "L2: no PMC pass of this kernel build -- algorithmic bytes (L2 / Infinity-Cache served) against the L2 -> L1 aggregate, not an HBM fraction"
        peak = L2_PEAK_GBS if l2_basis else HBM_PEAK_GBS
        roofline = {
            "bound": bound, "bound_note": bound_note, "achieved": hbm_rate if hbm_rate is not None else achieved, "peak": peak, "unit": "GB/s",
            "frac": (hbm_rate if hbm_rate is not None else achieved) / peak,
            "achieved_is": "HBM traffic (PMC FETCH_SIZE / WRITE_SIZE of this command) / kernel time" if hbm_rate is not None
                           else "ALGORITHMIC bytes / kernel time (no PMC profile of this configuration: not an HBM fraction)",
            "traffic": traffic, "algorithmic_GBps": achieved, "algorithmic_over_hbm_peak": achieved / HBM_PEAK_GBS,
            "kernel": kernel, "launches": scan_launches, "avg_launch_ms": avg_launch_s * 1e3,
            "algorithmic_bytes_per_launch": algo_bytes_per_launch,
            "hbm_frac": hbm_frac, "hbm_GBps": (traffic / avg_launch_s / 1e9) if traffic else None, "traffic_source": traffic_src,
            "one_pass_lower_bound_ms": index_pass_bytes / (HBM_PEAK_GBS * 1e9) * 1e3,
            "frac_of_one_pass_lower_bound": index_pass_bytes / (HBM_PEAK_GBS * 1e9) / avg_launch_s,
            "walk_adds_per_s": adds_per_s, "lds_add_u32_peak_per_s": LDS_ADD_U32_PEAK if info.postings_walk == 4 else LDS_ADD_U32_RANDOM,
            "frac_of_lds_add_peak": (adds_per_s / (LDS_ADD_U32_PEAK if info.postings_walk == 4 else LDS_ADD_U32_RANDOM)) if adds_per_s else None,
            "csr_scan_equivalent_GBps": csr_equiv, "bytes_per_csr_pass": info.bytes_per_pass, "merge_ms_total": merge_ms,
            "refine_ms_total": refine_ms, "exact_fallback_ms_total": fb_ms, "fallback_queries_last_step": info.last_fallbacks,
            "note": "algorithmic_GBps: bytes the dominant kernel has to read (the records of the batch's (query, column) posting lists + "
                    "their directory entries, vs_index_info.last_scan_bytes) / its measured time -- served mostly by L2 / Infinity Cache; "
                    "traffic / hbm_frac: PMC-measured HBM bytes of the same launch; one_pass_lower_bound: a single pass over the index at "
                    "8 TB/s (SURVEY 8(d), Qt = B); walk_adds: postings multiplied and scatter-added into LDS per second vs the measured "
                    "ds_add_u32 rate (tools/microbench/lds_scatter.hip) -- the resource this kernel is closest to",
        }
        line = {
            "metric": "queries/sec over 21M-doc sparse index, k=100; recall@100 vs reference",
            "value": qps, "unit": "queries/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.store == "fp32" else "f16 values, f32/f64 accumulate", "data": "synthetic",
            "config": {"workload": f"wiki21m-shaped sparse CSR index: {args.docs} docs x {NNZ_DOC} nnz, V={V}, {args.store} values, {args.columns} columns, "
                                   f"row-sharded over {world} GPU(s); {args.batch} queries/step ({NNZ_Q} nnz), k={args.k}",
                       "docs": args.docs, "docs_per_gpu": n_local, "batch": args.batch, "k": args.k, "queries_per_pass": qt, "columns": args.columns,
                       "lanes_per_row": info.lanes_per_row, "index_bytes_per_gpu": info.device_bytes,
                       "postings_copy_bytes_per_gpu": info.aux_bytes, "scan_path": path, "dominant_kernel": kernel, "index_build_s": round(build_s, 2)},
            "exchange": {"backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if world > 1 else None,
                         "process_group_backend": dist.get_backend() if world > 1 else None,
                         "nccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if world > 1 and backend == "nccl" else None,
                         "world_size": dist.get_world_size() if world > 1 else 1, "per_rank_ms_per_step": per_rank,
                         "collective": "one all_gather_into_tensor of B*k packed (score, global id) int64 per rank + vs_merge_topk" if world > 1 else None},
            "roofline": roofline,
        }
        line["roofline"]["kernel_source_hash"] = kernel_source_hash()
        # what the walk waits for: the L1 -> L2 request rate, the L2 hit rate, and the rate as a fraction of the request
        # ceiling at that hit rate -- from the committed utilisation passes of THIS kernel build (null otherwise), like the traffic
        ut = pmc_utilisation("bp_quad_topk" if info.postings_walk == 4 else "bp_bq_topk" if info.postings_walk == 6 else "bp_walk_topk") if info.last_path == 3 else None
        for key in ("l2_requests_per_clk_cu", "l2_hit_rate", "request_ceiling_frac"):
            line["roofline"][key] = (ut or {}).get(key)
        line["roofline"]["utilisation_source"] = (f"profiles/pmc_summary.json utilisation ({ut.get('tag')}: rocprofv3 --pmc passes, {ut.get('shape')}); ceiling: "
                                                  "tools/microbench/l2_requests.hip (0.40 requests a clock and CU from L2, 0.095 beyond)") if ut else None
        if world > 1:
            line["parity"] = sharded_parity
        if world == 1:
            line["parity"] = parity_check(local_rank, kind, expect_path=info.last_path if n_local >= 20_000 and args.scan == "auto" else None)
            if not args.no_secondary and args.docs == N_DOCS and args.columns == "uniform" and args.scan == "auto":
                line["secondary"] = secondary(index, batches, args, local_rank, device, elapsed / args.steps)
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(args.cpu_sample_docs, local_rank, kind)
        line["roofline"]["cycles_per_chunk_and_cu"] = (avg_launch_s * SHADER_CLOCK_HZ * 256 / (algo_bytes_per_launch / 256.0)) if info.postings_walk == 4 and algo_bytes_per_launch > 0 else None
        line["summary"] = summarise(line)                     # LAST key: the driver keeps the line's tail
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
