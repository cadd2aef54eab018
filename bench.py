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
LDS_ADD_U32_PEAK = 5.0e12    # ds_add_u32 at random addresses, all 256 CUs (profiles/r02_lds_scatter.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--docs", type=int, default=N_DOCS, help="total index rows (default: the metric's 21 015 324)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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


def parity_check(device, kind=0):
    """Small prefix of the same synthetic index (rows are a pure function of (seed, row id)) searched
    by the HIP path and by the CPU oracle: recall@100 and max relative score error."""
    import oracle
    from oracle import compare
    from vsearch_amd.device_index import DeviceIndex
    n = 20_000
    idx = DeviceIndex.synthetic(INDEX_SEED, 0, n, V, NNZ_DOC, kind, 0, 0, device)
    q = oracle.synth_queries(QUERY_SEED, 8, V, NNZ_Q, kind=kind)
    ids, sc = idx.search(q, K)
    ip, ix, d = idx.export_csr()
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix.astype(np.int32), d, V, q, K, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=1e-4)
    rel = float(np.max(np.abs(sc.astype(np.float64) - o_sc) / np.abs(o_sc)))
    return {"docs": n, "queries": 8, "recall_at_100_vs_oracle": compare.recall_at_k(o_ids, ids), "max_rel_score_err": rel}


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
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ids, scores = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    Profile.enable(False)
    scan_ms, scan_launches = Profile.read("csr_scan_topk")
    merge_ms, _ = Profile.read("merge_topk")
    refine_ms, _ = Profile.read("refine_topk")
    fb_ms, _ = Profile.read("exact_fallback")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
                3: "blocked postings: int32 fixed-point filter walk (8 queries per tile) + exact refine of k+28 candidates"}[info.last_path]
        kernel = {0: "csr_scan_topk_wave", 1: "csr_scan_topk_mq", 2: "bp_walk_topk", 3: "bp_walk_topk"}[info.last_path]
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
                        and rec.get("store", "fp32") == args.store and rec.get("scan", "auto") == args.scan and rec.get("columns", "uniform") == args.columns:
                    traffic = rec["hbm_bytes_per_launch"]
                    traffic_src = f"profiles/pmc_summary.json ({rec.get('tag', '?')}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command)"
            except Exception:
                traffic = None
        index_pass_bytes = info.bytes_per_pass                     # one pass over the shard's CSR packets (SURVEY 8(d) bytes_pass)
        hbm_frac = (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None
        adds_per_s = info.last_walk_postings / launches_per_step / avg_launch_s if info.last_path >= 2 else None
        if info.last_path >= 2:
            bound = "hbm" if (hbm_frac or 0.0) > 0.5 else "on-chip: LDS scatter-adds into random document slots (bank conflicts; ds_add_u32) + the L2->L1 record loads they overlap with"
        else:
            bound = "hbm"
        # `achieved` / `frac`: the HBM rate the counters evidence when a matching PMC profile exists (the honest HBM fraction);
        # without one, the algorithmic rate (bytes the kernel has to read / time), flagged as such -- on the postings path most of
        # those bytes come from L2 / Infinity Cache, so that rate can exceed the HBM peak and says nothing about HBM utilisation.
        hbm_rate = (traffic / avg_launch_s / 1e9) if traffic else None
        roofline = {
            "bound": bound, "achieved": hbm_rate if hbm_rate is not None else achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (hbm_rate if hbm_rate is not None else achieved) / HBM_PEAK_GBS,
            "achieved_is": "HBM traffic (PMC FETCH_SIZE / WRITE_SIZE of this command) / kernel time" if hbm_rate is not None
                           else "ALGORITHMIC bytes / kernel time (no PMC profile of this configuration: not an HBM fraction)",
            "traffic": traffic, "algorithmic_GBps": achieved, "algorithmic_over_hbm_peak": achieved / HBM_PEAK_GBS,
            "kernel": kernel, "launches": scan_launches, "avg_launch_ms": avg_launch_s * 1e3,
            "algorithmic_bytes_per_launch": algo_bytes_per_launch,
            "hbm_frac": hbm_frac, "hbm_GBps": (traffic / avg_launch_s / 1e9) if traffic else None, "traffic_source": traffic_src,
            "one_pass_lower_bound_ms": index_pass_bytes / (HBM_PEAK_GBS * 1e9) * 1e3,
            "frac_of_one_pass_lower_bound": index_pass_bytes / (HBM_PEAK_GBS * 1e9) / avg_launch_s,
            "walk_adds_per_s": adds_per_s, "lds_add_u32_peak_per_s": LDS_ADD_U32_PEAK,
            "frac_of_lds_add_peak": (adds_per_s / LDS_ADD_U32_PEAK) if adds_per_s else None,
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
                         "world_size": dist.get_world_size() if world > 1 else 1,
                         "collective": "one all_gather_into_tensor of B*k packed (score, global id) int64 per rank + vs_merge_topk" if world > 1 else None},
            "roofline": roofline,
        }
        if world == 1:
            line["parity"] = parity_check(local_rank, kind)
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(args.cpu_sample_docs, local_rank, kind)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
