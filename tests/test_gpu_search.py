"""GPU parity tests of the search hot path (index.py:88-94) through the C ABI -- run on MI355X.

HIP path vs (a) the committed golden vectors captured from the reference and (b) the CPU oracle on
the same seeded inputs.  Bars: binary index x dyadic queries bit-exact; fp32 paths 1e-4 relative
(north_star), ids identical modulo (near-)ties.
"""
import numpy as np
import pytest

import oracle
from oracle import compare
from conftest import V
from vsearch_amd import synth
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, ShardGroup, merge_topk

pytestmark = pytest.mark.gpu

RTOL = 1e-4   # fp32 score tolerance stated by BASELINE.json north_star


@pytest.fixture(scope="module")
def sparse2000():
    ip, ix, d = oracle.synth_csr(0, 0, 2000)
    return ip, ix, d, DeviceIndex.from_csr(ip, ix, d, V)


def test_library_reports_gfx950_device():
    assert nat.device_count() >= 1
    assert nat.lib().vs_version() >= 100


@pytest.mark.parametrize("qt", [0, 1], ids=["multi-query", "dense-image"])
@pytest.mark.parametrize("k", [1, 100, 128, 129, 1024, 2000])
def test_sparse_fp32_vs_golden_and_oracle(golden, sparse2000, k, qt):
    ip, ix, d, idx = sparse2000
    g = golden("search_sparse_n2000")
    q = oracle.synth_queries(1, 8)
    idx.set_queries_per_pass(qt)
    ids, sc = idx.search(q, k)
    idx.set_queries_per_pass(0)
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q, k, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL)
    if f"ids_k{k}" in g.files:
        compare.compare_topk(g[f"ids_k{k}"], g[f"scores_k{k}"], ids, sc, rtol=RTOL)
        assert compare.recall_at_k(g[f"ids_k{k}"], ids) >= 0.999


def test_sparse_all_scores_match_oracle(sparse2000):
    ip, ix, d, idx = sparse2000
    q = oracle.synth_queries(1, 8)
    got = idx.scores(q)
    _, _, want = oracle.csr_search(ip, ix, d, V, q, 1, acc64=True, return_all=True)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("qt", [0, 1], ids=["multi-query", "dense-image"])
def test_sparse_n20000_golden(golden, qt):
    g = golden("search_sparse_n20000")
    ip, ix, d = oracle.synth_csr(0, 0, 20000)
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    idx.set_queries_per_pass(qt)
    q = oracle.synth_queries(1, 32)
    ids, sc = idx.search(q, 100)
    assert idx.info().queries_per_pass == (8 if qt == 0 else 1)
    compare.compare_topk(g["ids_k100"], g["scores_k100"], ids, sc, rtol=RTOL)
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q, 100, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL)


def test_multi_query_tiles_ragged_batches_and_dense_fallback():
    """Tile planning: batch sizes that are not multiples of 8, queries of very different density, and a
    query too dense for the LDS tile tables (whole batch falls back to the dense-image pass)."""
    n = 3000
    ip, ix, d = oracle.synth_csr(21, 0, n, V, 200)
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    rng = np.random.default_rng(1)
    q = np.zeros((13, V), np.float32)
    for i, nz in enumerate([1, 5, 776, 2000, 40, 776, 3000, 9, 776, 776, 100, 2, 1500]):
        cols = rng.choice(V, size=nz, replace=False)
        q[i, cols] = rng.uniform(-1, 2, size=nz).astype(np.float32)
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q, 64, acc64=True, return_all=True)
    ids, sc = idx.search(q, 64)
    assert idx.info().queries_per_pass == 8
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL, tie_rtol=1e-5)
    q[4] = rng.uniform(0, 1, size=V).astype(np.float32)              # fully dense query
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q, 64, acc64=True, return_all=True)
    ids, sc = idx.search(q, 64)
    assert idx.info().queries_per_pass == 1
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    empty = np.zeros((3, V), np.float32)                               # all-zero queries: every score 0, lowest ids win
    ids, sc = idx.search(empty, 5)
    assert (ids == np.arange(5)).all() and (sc == 0).all()


def test_k_gt_n_raises_like_topk(sparse2000):
    *_, idx = sparse2000
    with pytest.raises(RuntimeError, match="out of range"):
        idx.search(oracle.synth_queries(1, 2), 2001)


@pytest.mark.parametrize("qt", [0, 1], ids=["multi-query", "dense-image"])
def test_multipass_large_k(qt):
    """k beyond one pass's capacity (512 ranks for the Qt = 8 pass, 2048 for Qt = 1): repeated passes with an exclusive
    upper-bound key ("search after"); result = full canonical ranking."""
    n = 5000
    ip, ix, d = oracle.synth_csr(2, 0, n, V, 64)
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    idx.set_queries_per_pass(qt)
    q = oracle.synth_queries(6, 11)
    _, _, allsc = oracle.csr_search(ip, ix, d, V, q, 1, acc64=True, return_all=True)
    for k in (513, 1300, n):
        ids, sc = idx.search(q, k)
        assert idx.info().queries_per_pass == (8 if qt == 0 else 1)
        compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
        assert all(len(set(r.tolist())) == k for r in ids)
        o_ids, o_sc = oracle.csr_search(ip, ix, d, V, q, k, acc64=True)
        compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL)


@pytest.mark.parametrize("tag,exact", [("f32", False), ("dyadic", True)])
def test_bot_binary_index(golden, tag, exact):
    g = golden("search_bot")
    n, b = int(g["n"]), int(g["b"])
    ip, ix, _ = oracle.synth_csr(int(g["index_seed"]), 0, n, V, int(g["nnz"]), synth.KIND_BOT)
    idx = DeviceIndex.from_csr(ip, ix, None, V)
    assert idx.info().store_dtype == nat.VS_NONE
    seed = int(g["query_seeds"][1 if exact else 0])
    q = oracle.synth_queries(seed, b, val_law=synth.VAL_DYADIC if exact else synth.VAL_GRID)
    for k, qt in ((10, 0), (100, 0), (100, 1)):
        idx.set_queries_per_pass(qt)
        ids, sc = idx.search(q, k)
        compare.compare_topk(g[f"{tag}_ids_k{k}"], g[f"{tag}_scores_k{k}"], ids, sc, rtol=RTOL, exact=exact)
        o_ids, o_sc, allsc = oracle.csr_search(ip, ix, None, V, q, k, return_all=True)
        compare.check_topk_valid(allsc, ids, sc, rtol=RTOL, exact=exact, canonical=exact)
        if exact:       # bit-exact scores AND identical ids in canonical order
            assert (ids == o_ids).all() and (sc == o_sc).all()
    if exact:
        assert (idx.scores(q) == allsc).all()


def test_fp16_index_rounds_query_and_values():
    """fp16 store = the reference's `fp16=True` load default (index.py:135,176) + q.type(fp16) (index.py:89)."""
    n = 3000
    ip, ix, d = oracle.synth_csr(4, 0, n)
    idx = DeviceIndex.from_csr(ip, ix, d, V, store_dtype=nat.VS_F16)
    q = oracle.synth_queries(5, 4)
    ids, sc = idx.search(q, 50)
    d16 = d.astype(np.float16).astype(np.float32)
    q16 = q.astype(np.float16).astype(np.float32)
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d16, V, q16, 50, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL)


def test_export_roundtrip_and_ragged_rows():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 40, size=300)
    lens[[0, 17, 299]] = 0                                   # empty rows, incl. first and last
    lens[5] = 1000
    ip = np.zeros(301, np.int64)
    np.cumsum(lens, out=ip[1:])
    ix = np.concatenate([np.sort(rng.choice(V, size=l, replace=False)) for l in lens]).astype(np.int32)
    d = rng.uniform(-2, 2, size=ix.size).astype(np.float32)
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    e_ip, e_ix, e_d = idx.export_csr()
    assert (e_ip == ip).all() and (e_ix == ix).all() and (e_d == d).all()
    q = rng.uniform(-1, 1, size=(5, V)).astype(np.float32)   # dense query with negative weights
    ids, sc = idx.search(q, 300)
    _, _, allsc = oracle.csr_search(ip, ix, d, V, q, 1, acc64=True, return_all=True)
    np.testing.assert_allclose(idx.scores(q), allsc, rtol=1e-5, atol=1e-5)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    # int64 indices / int32 indptr variants
    idx2 = DeviceIndex.from_csr(ip.astype(np.int32), ix.astype(np.int64), d, V)
    assert (idx2.export_csr()[1] == ix).all()


def test_create_rejects_bad_input():
    ip = np.array([0, 2], np.int64)
    with pytest.raises(ValueError):
        DeviceIndex.from_csr(ip, np.array([0, V], np.int32), np.ones(2, np.float32), V)     # column out of range
    with pytest.raises(NotImplementedError):
        DeviceIndex.from_csr(ip, np.array([0, 1], np.int32), np.ones(2, np.float32), 70000)  # uint16 column ids
    with pytest.raises(ValueError):
        DeviceIndex.from_csr(ip, np.array([0, 1], np.int32), np.array([1, 2], np.float32), V, store_dtype=nat.VS_NONE)


@pytest.mark.parametrize("kind,nnz,law,store", [(synth.KIND_VDR, 768, synth.VAL_GRID, nat.VS_F32),
                                                (synth.KIND_BOT, 86, synth.VAL_ONE, nat.VS_NONE),
                                                (synth.KIND_VDR, 100, synth.VAL_DYADIC, nat.VS_F32)])
def test_device_synth_matches_host_twins(kind, nnz, law, store):
    idx = DeviceIndex.synthetic(9, 12345, 700, V, nnz, kind, law, store)
    ip, ix, d = oracle.synth_csr(9, 12345, 700, V, nnz, kind, law)
    e_ip, e_ix, e_d = idx.export_csr()
    assert (e_ip == ip).all() and (e_ix == ix).all() and (e_d == d).all()


def test_dense_index_vs_golden(golden):
    g = golden("search_dense")
    n, b = int(g["n"]), int(g["b"])
    ip, ix, d = oracle.synth_csr(0, 0, n)
    dense = np.zeros((n, V), np.float32)
    dense[np.repeat(np.arange(n), 768), ix] = d
    idx = DeviceIndex.from_dense(dense)
    q = oracle.synth_queries(1, b)
    for k in (1, 100):
        ids, sc = idx.search(q, k)
        compare.compare_topk(g[f"ids_k{k}"], g[f"scores_k{k}"], ids, sc, rtol=RTOL)
    n2, b2, s1, s2 = g["full_shape"].tolist()
    m2, q2 = synth.dense_uniform(s1, (n2, V), 0.0, 1.0), synth.dense_uniform(s2, (b2, V), 0.0, 1.0)
    idx2 = DeviceIndex.from_dense(m2)
    ids, sc = idx2.search(q2, 50)
    # 29 523-term fp32 dot products: summation order moves scores by ~1e-6 relative (reference GEMM vs ours)
    compare.compare_topk(g["full_ids_k50"], g["full_scores_k50"], ids, sc, rtol=RTOL, tie_rtol=5e-6)
    want = q2.astype(np.float64) @ m2.astype(np.float64).T
    np.testing.assert_allclose(idx2.scores(q2), want, rtol=2e-5)
    assert (idx2.export_dense() == m2).all()
    with pytest.raises(RuntimeError):
        idx2.search(q2, n2 + 1)


def test_merge_topk_matches_oracle():
    rng = np.random.default_rng(3)
    B, n, k = 7, 800, 100
    ids = np.stack([rng.permutation(100000)[:n] for _ in range(B)]).astype(np.int64)
    sc = rng.integers(0, 50, size=(B, n)).astype(np.float32) / 8      # many exact ties
    got = merge_topk(ids, sc, k)
    want = oracle.merge_topk(ids, sc, k)
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all()


def test_torch_device_tensors_roundtrip(sparse2000):
    import torch
    ip, ix, d, idx = sparse2000
    q = torch.from_numpy(oracle.synth_queries(1, 8)).cuda()
    ids, sc = idx.search(q, 100)
    assert ids.is_cuda and ids.dtype == torch.int64 and sc.dtype == torch.float32
    ids_h, sc_h = idx.search(q.cpu().numpy(), 100)
    assert (ids.cpu().numpy() == ids_h).all() and (sc.cpu().numpy() == sc_h).all()
    qh = q.half()                                            # fp16 query input on an fp32 index
    ids2, sc2 = idx.search(qh, 10)
    o_ids, o_sc = oracle.csr_search(ip, ix, d, V, qh.float().cpu().numpy(), 10, acc64=True)
    compare.compare_topk(o_ids, o_sc, ids2.cpu().numpy(), sc2.cpu().numpy(), rtol=RTOL)


# ---- BASELINE-scale properties (sizes the oracle cannot score in seconds) ------------------------------
def test_large_index_topk_is_valid_and_shards_merge_exactly():
    """2 M synthetic docs (9.2 GB): (1) the fused top-k equals a top-k of the independently computed dense
    score matrix (csr_scan_scores kernel), (2) searching two row shards and merging == searching the whole."""
    n, b, k = 2_000_000, 4, 100
    whole = DeviceIndex.synthetic(0, 0, n)
    q = oracle.synth_queries(1, b)
    ids, sc = whole.search(q, k)
    allsc = whole.scores(q)                                   # Qt = 1 scores-only kernel: independent code path
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    assert (np.diff(sc.astype(np.float64), axis=1) <= 0).all()
    half = n // 2
    lo, hi = DeviceIndex.synthetic(0, 0, half), DeviceIndex.synthetic(0, half, n - half)
    i0, s0 = lo.search(q, k, id_offset=0)
    i1, s1 = hi.search(q, k, id_offset=half)
    m_ids, m_sc = merge_topk(np.concatenate([i0, i1], 1), np.concatenate([s0, s1], 1), k)
    assert (m_ids == ids).all() and (m_sc == sc).all()
    # idempotence: same call, same bits (reproducible accumulation order)
    ids2, sc2 = whole.search(q, k)
    assert (ids2 == ids).all() and (sc2 == sc).all()


def test_large_bot_index_exact_against_scores_kernel():
    n, b, k = 3_000_000, 8, 100
    idx = DeviceIndex.synthetic(3, 0, n, V, 86, synth.KIND_BOT, 0, nat.VS_NONE)
    q = oracle.synth_queries(5, b, val_law=synth.VAL_DYADIC)
    ids, sc = idx.search(q, k)
    allsc = idx.scores(q)
    compare.check_topk_valid(allsc, ids, sc, exact=True, canonical=True)   # bit-exact scores, canonical ids


def test_nccl_single_rank_exchange_path():
    """The all-gather + device merge of the row-sharded search, exercised over RCCL with a 1-rank group."""
    import os
    import torch
    import torch.distributed as dist
    from vsearch_amd.distributed import ShardedSearcher
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n = 5000
        ip, ix, d = oracle.synth_csr(0, 0, n)
        idx = DeviceIndex.from_csr(ip, ix, d, V)
        s = ShardedSearcher.from_device_index(idx, 0, n)
        s.force_exchange = True
        q = torch.from_numpy(oracle.synth_queries(1, 6)).cuda()
        ids, sc = s.search(q, 100)
        ref_ids, ref_sc = idx.search(q, 100)
        assert ids.is_cuda and (ids == ref_ids).all() and (sc == ref_sc).all()
    finally:
        dist.destroy_process_group()


def test_two_rank_bench_launch_on_one_gpu_equals_the_unsharded_search(tmp_path):
    """the multi-rank path end to end on the hardware at hand -- `bench.py --gpus 2` through launch.spawn_ranks
    (two fresh rank processes, env:// rendezvous on 127.0.0.1; both share cuda:0 and exchange over gloo because RCCL refuses two
    ranks on one device), row shards of 200 000 docs each, one all-gather, merge.  Rank 0's ids and scores must be the unsharded
    index's, bit for bit; the JSON line carries the per-rank phase breakdown."""
    import json, os, subprocess, sys
    from vsearch_amd import launch
    import bench
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "ids.npz")
    out = str(tmp_path / "line.json")
    docs, batch, k = 400_000, 64, 100
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--docs", str(docs), "--batch", str(batch), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--dump-ids", dump]
    env = dict(os.environ, VS_BENCH_SHARE_GPU="1")
    # the parent of the ranks must not have touched the GPU: run the launcher in a fresh interpreter (this pytest process has)
    code = ("import sys, json; sys.path.insert(0, %r); from vsearch_amd import launch; "
            "raise SystemExit(launch.spawn_ranks(%r, 2, timeout_s=600, extra_env={'VS_BENCH_SHARE_GPU': '1'}))" % (repo, cmd))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["exchange"]["world_size"] == 2 and line["exchange"]["process_group_backend"] == "gloo"
    pr = line["exchange"]["per_rank_ms_per_step"]
    assert len(pr["by_rank"]) == 2 and pr["walk_ms"]["max"] > 0 and pr["exchange_ms"]["max"] > 0 and pr["merge_ms"]["max"] > 0
    par = line["parity"]                                             # the oracle check through the sharded path (all ranks search, rank 0 compares)
    assert par["ranks"] == 2 and par["recall_at_100_vs_oracle"] == 1.0 and par["max_rel_score_err"] < 1e-4
    got = np.load(dump)
    idx = DeviceIndex.synthetic(bench.INDEX_SEED, 0, docs, V, bench.NNZ_DOC, 0, 0, nat.VS_F32)
    import torch
    qb = bench.make_query_batches(3, batch, torch.device("cuda", 0))
    ids, sc = idx.search(qb[2], k)                                   # steps 2 + warmup 1: the last step searched batch 2
    assert (ids.cpu().numpy() == got["ids"]).all() and (sc.cpu().numpy() == got["scores"]).all()


def test_eight_rank_bench_launch_on_one_gpu_equals_the_unsharded_search(tmp_path):
    """eight rank processes sharing the one GPU (gloo exchange), 20 000-doc shards -- every rank's shard is big
    enough for the postings filter, the path the measured run takes (bench.parity_sharded asserts it) -- rank 0's ids and scores equal
    the unsharded index's bit for bit."""
    import json, os, subprocess, sys
    import bench
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "ids8.npz")
    docs, batch, k, world = 8 * 20_000, 32, 100, 8
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(world), "--docs", str(docs), "--batch", str(batch), "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--dump-ids", dump]
    env = dict(os.environ, VS_BENCH_SHARE_GPU="1")
    code = ("import sys, json; sys.path.insert(0, %r); from vsearch_amd import launch; "
            "raise SystemExit(launch.spawn_ranks(%r, %d, timeout_s=900, extra_env={'VS_BENCH_SHARE_GPU': '1'}))" % (repo, cmd, world))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["exchange"]["world_size"] == world
    par = line["parity"]
    assert par["ranks"] == world and par["docs_per_rank"] == 20_000 and par["scan_path_rank0"] == 3
    assert par["recall_at_100_vs_oracle"] == 1.0 and par["max_rel_score_err"] < 1e-4
    got = np.load(dump)
    idx = DeviceIndex.synthetic(bench.INDEX_SEED, 0, docs, V, bench.NNZ_DOC, 0, 0, nat.VS_F32)
    import torch
    qb = bench.make_query_batches(2, batch, torch.device("cuda", 0))
    ids, sc = idx.search(qb[1], k)                                   # steps 1 + warmup 1: the last step searched batch 1
    assert (ids.cpu().numpy() == got["ids"]).all() and (sc.cpu().numpy() == got["scores"]).all()


@pytest.mark.parametrize("kind", ["plain", "bot", "zipf"])
def test_four_processes_sharing_the_gpu_keep_their_results(kind):
    """Four processes on the one GPU, 120 searches each, every result against the CSR scan (tools/contention_check.py): other processes'
    waves on a CU stretch the timing windows inside a workgroup -- the run that exposed a quad walk which lost whole blocks of candidates
    in 10 - 25 % of the searches while every single-process test passed (round 5; root cause, round 6: the epilogue's cut decision read
    per thread from counters the next round was already pushing to -- test_a_late_wave_does_not_split_the_cut_decision forces that window
    in a single process).  The quad chunks, the bag-of-token chunks and the head pre-pass + list walk."""
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "contention_check.py"), "4", "120", kind], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if "bad searches" in l]
    assert r.returncode == 0 and len(lines) == 4 and all(": 0 bad searches" in l and "path 3" in l for l in lines), r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("kind", ["dense", "mask"])
def test_four_processes_sharing_the_gpu_dense_and_mask_kernels(kind):
    """The same rig on the dense MFMA search and on the encoder's mask kernels: four processes, every result against
    torch.  The fused mask -> CSR kernel is the one kernel here whose workgroups wait for each other (a workgroup places its rows once
    the workgroups with lower tickets have published their totals): with four processes on the GPU they are not all resident at once."""
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "contention_check.py"), "4", "60", kind], capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if "bad searches" in l]
    assert r.returncode == 0 and len(lines) == 4 and all(": 0 bad searches" in l for l in lines), r.stdout[-2000:] + r.stderr[-1000:]


def test_reserve_and_append_equals_single_shot():
    """Shard-by-shard construction (vs_index_create_reserved + vs_index_append_csr) == one-shot creation."""
    n = 1500
    ip, ix, d = oracle.synth_csr(31, 0, n, V, 300)
    whole = DeviceIndex.from_csr(ip, ix, d, V)
    cuts = [0, 1, 400, 401, 1100, n]
    packets = int(((np.diff(ip) + 7) // 8).sum())
    inc = DeviceIndex.reserved(n + 10, packets + 5, V, nat.VS_F32)
    for a, b in zip(cuts[:-1], cuts[1:]):
        inc.append_csr(ip[a:b + 1] - ip[a], ix[ip[a]:ip[b]], d[ip[a]:ip[b]])
    assert inc.info().n_rows == n and inc.info().nnz == ip[-1]
    for x, y in zip(inc.export_csr(), whole.export_csr()):
        assert (x == y).all()
    q = oracle.synth_queries(2, 5)
    for x, y in zip(inc.search(q, 50), whole.search(q, 50)):
        assert (x == y).all()
    with pytest.raises(ValueError, match="reserved"):
        inc.append_csr(ip[:12] - ip[0], ix[:ip[11]], d[:ip[11]])          # 11 more rows than reserved


def test_sparsity_aware_dense_index(golden):
    """A dense index that is > 95 % zeros (VDR embeddings kept dense, retriever.py:292-297) is stored as CSR packets
    and searched by the CSR scan: same results as the MFMA path and the golden; a really dense matrix stays dense."""
    g = golden("search_dense")
    n, b = int(g["n"]), int(g["b"])
    ip, ix, d = oracle.synth_csr(0, 0, n)
    dense = np.zeros((n, V), np.float32)
    dense[np.repeat(np.arange(n), 768), ix] = d
    auto = DeviceIndex.from_dense(dense, max_density=0.05)
    info = auto.info()
    assert info.kind == nat.VS_KIND_DENSE and info.n_packets == n * 96 and info.nnz == n * 768
    q = oracle.synth_queries(1, b)
    for k in (1, 100):
        ids, sc = auto.search(q, k)
        compare.compare_topk(g[f"ids_k{k}"], g[f"scores_k{k}"], ids, sc, rtol=RTOL)
    mfma = DeviceIndex.from_dense(dense)                                   # max_density = 0: dense kernel
    assert mfma.info().n_packets == 0
    i2, s2 = mfma.search(q, 100)
    compare.compare_topk(i2, s2, *auto.search(q, 100), rtol=RTOL)
    assert (auto.export_dense() == dense).all()
    h16 = DeviceIndex.from_dense(dense.astype(np.float16), max_density=0.05)
    assert (h16.export_dense(np.float16) == dense.astype(np.float16)).all()
    full = synth.dense_uniform(41, (64, V), 0.0, 1.0)
    assert DeviceIndex.from_dense(full, max_density=0.05).info().n_packets == 0


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_against_oracle(seed):
    """Randomised small problems: odd vocabulary sizes, ragged/empty rows, B and k not multiples of anything,
    all three value modes, both scan families -- each checked against the full oracle score matrix."""
    rng = np.random.default_rng(100 + seed)
    n_cols = int(rng.choice([7, 40, 257, 1000, 4099, 65535]))
    n = int(rng.integers(1, 700))
    max_len = min(n_cols, int(rng.choice([1, 5, 60, 300])))
    lens = rng.integers(0, max_len + 1, size=n)
    ip = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=ip[1:])
    ix = np.concatenate([np.sort(rng.choice(n_cols, size=l, replace=False)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    mode = seed % 3
    d = None if mode == 2 else rng.integers(1, 200, size=ix.size).astype(np.float32) / 16          # exactly representable in fp16 too
    B = int(rng.integers(1, 20))
    q = np.zeros((B, n_cols), np.float32)
    for i in range(B):
        nz = int(rng.integers(0, min(n_cols, 50) + 1))
        q[i, rng.choice(n_cols, size=nz, replace=False)] = rng.integers(-64, 128, size=nz).astype(np.float32) / 32
    idx = DeviceIndex.from_csr(ip, ix, d, n_cols, store_dtype={0: nat.VS_F32, 1: nat.VS_F16, 2: nat.VS_NONE}[mode])
    _, _, allsc = oracle.csr_search(ip, ix, d, n_cols, q, 1, acc64=True, return_all=True)
    assert (idx.scores(q) == allsc).all()                      # small dyadic values: every sum is exact
    for k in sorted({1, min(n, 7), min(n, 130), n}):
        for qt in (0, 1):
            idx.set_queries_per_pass(qt)
            ids, sc = idx.search(q, k)
            compare.check_topk_valid(allsc, ids, sc, exact=True, canonical=True)


def test_dense_index_large_k_multipass():
    """Dense index, k > 2048 (slices below an exclusive upper-bound key) and 4096 < N (radix-select path) with ties."""
    rng = np.random.default_rng(5)
    n, v = 9000, 96
    mat = (rng.integers(0, 8, size=(n, v)).astype(np.float32)) / 4            # many tied scores
    q = (rng.integers(0, 4, size=(3, v)).astype(np.float32)) / 2
    idx = DeviceIndex.from_dense(mat)
    want = q.astype(np.float64) @ mat.astype(np.float64).T
    for k in (100, 2048, 2049, 5000, n):
        ids, sc = idx.search(q, k)
        compare.check_topk_valid(want.astype(np.float32), ids, sc, exact=True, canonical=True)


def test_dense_index_tail_split_along_k():
    """Dense index whose documents leave a few 128-document tiles after the full rounds of the grid (66 000 docs, B = 130: 1024
    tiles in two rounds + 8): those are split along K over the idle slots and their partial sums added in slice order
    (launch_dense_scores).  Scores of main and tail documents against an fp64 matmul; the integer-valued matrix makes every sum
    exact, so ids and scores must be THE top-k whatever the summation order."""
    import torch
    n, v, b = 66_000, 2048, 130
    g = torch.Generator(device="cuda").manual_seed(11)
    mat = torch.randint(0, 8, (n, v), device="cuda", generator=g).float() / 4
    q = torch.randint(0, 4, (b, v), device="cuda", generator=g).float() / 2
    idx = DeviceIndex.from_dense(mat)
    want = (q.double() @ mat.double().t()).float().cpu().numpy()
    for k in (1, 100):
        ids, sc = idx.search(q, k)
        compare.check_topk_valid(want, ids.cpu().numpy(), sc.cpu().numpy(), exact=True, canonical=True)
    allsc = idx.scores(q)
    assert (np.asarray(allsc.cpu() if hasattr(allsc, "cpu") else allsc) == want).all()


def test_dense_scores_do_not_depend_on_tail_position_or_batch_size():
    """the split-K tail must add a document's 512-column block sums in the main kernel's order.  Non-dyadic values (every
    fp32 sum rounds): rows copied into the tail of the grid score bit-identically to their originals in a main round, and a batch of
    128 queries scores bit-identically to the same queries inside a batch of 256 (another tile plan)."""
    import torch
    n_main, n_copy, v = 65_536, 700, 2048
    g = torch.Generator(device="cuda").manual_seed(5)
    mat = torch.rand((n_main + n_copy, v), device="cuda", generator=g) * 3 + 0.01
    mat[n_main:] = mat[:n_copy]                                        # the tail documents are copies of main-round documents
    q = torch.rand((256, v), device="cuda", generator=g) * 3 + 0.01
    idx = DeviceIndex.from_dense(mat)
    s256 = idx.scores(q)
    s256 = s256.cpu().numpy() if hasattr(s256, "cpu") else np.asarray(s256)
    assert (s256[:, n_main:] == s256[:, :n_copy]).all(), "a tail document scores differently from its copy in a main round"
    s128 = idx.scores(q[:128])
    s128 = s128.cpu().numpy() if hasattr(s128, "cpu") else np.asarray(s128)
    assert (s128 == s256[:128]).all(), "scores depend on the batch size"
    want = (q.double() @ mat.double().t()).cpu().numpy()
    assert np.abs(s256 - want).max() / np.abs(want).max() < 1e-5


def _zipf_csr(rng, n, nnz, s, perm, binary=False, dyadic=False):
    """Rows with Zipf(s) column popularity, distinct sorted columns (numpy; the popular columns end up in most rows)."""
    w = 1.0 / np.arange(1, V + 1) ** s
    keys = rng.random((n, V)) ** (1.0 / w)                       # Efraimidis-Spirakis weighted sampling without replacement
    cols = np.sort(perm[np.argpartition(-keys, nnz, axis=1)[:, :nnz]], axis=1).astype(np.int32)
    ip = np.arange(0, (n + 1) * nnz, nnz, dtype=np.int64)
    if dyadic:
        vals = rng.integers(1, 256, size=cols.shape).astype(np.float32) / 64
    else:
        vals = (0.01 + 3 * rng.random(cols.shape)).astype(np.float32)
    return ip, cols.reshape(-1), (None if binary else vals.reshape(-1))


@pytest.mark.parametrize("binary", [False, True], ids=["fp32", "binary-dyadic"])
@pytest.mark.parametrize("s", [0.75, 1.2])
def test_skewed_column_popularity_shared_column_variant(s, binary):
    """Head columns shared by most queries AND most documents (SURVEY §8(d) 'Zipf column popularity' run): the
    multi-query pass switches to its shared-column variant; both variants must agree with the oracle."""
    rng = np.random.default_rng(5)
    perm = rng.permutation(V)
    n, B = 1500, 21
    ip, ix, d = _zipf_csr(rng, n, 86 if binary else 300, s, perm, binary=binary)
    qip, qix, qd = _zipf_csr(rng, B, 400, s, perm, dyadic=binary)
    q = np.zeros((B, V), dtype=np.float32)
    q[np.repeat(np.arange(B), 400), qix] = qd
    idx = DeviceIndex.from_csr(ip, ix, d, V)
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q, 50, acc64=True, return_all=True)
    res = {}
    idx.set_option("blocked_postings", 0)
    for forced in ("0", "1", None):                               # kernel variants: pairs, shared columns; auto
        idx.set_option("mq_variant", -1 if forced is None else int(forced))
        ids, sc = idx.search(q, 50)
        assert idx.info().queries_per_pass == 8
        res[forced] = (ids, sc)
        if binary:                                                # integer path: bit-exact scores, canonical ids
            assert (sc == o_sc).all() and (ids == o_ids).all()
        else:
            compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
            compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL)
    np.testing.assert_allclose(res["0"][1], res["1"][1], rtol=1e-6)


@pytest.mark.parametrize("B,k", [(1, 100), (3, 17), (2, 400)])
def test_small_batch_many_chunks_merge_prefilter(B, k):
    """A small batch cuts the rows into up to one chunk per CU; the merge of the per-chunk (sorted) top-k lists takes the
    sorted-run shortcut.  Adversarial layout: every top document sits in the first chunk."""
    rng = np.random.default_rng(11)
    n, nnz = 150_000, 16
    q = oracle.synth_queries(1, B)
    cols = np.sort(rng.integers(0, V, size=(n, nnz)).astype(np.int32), axis=1)
    hot = np.nonzero(q[0])[0][:nnz].astype(np.int32)               # rows 0..599 hit query 0's columns with growing weights
    cols[:600] = np.sort(hot)
    keep = np.ones((n, nnz), dtype=bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]                       # drop duplicate columns in a row
    ip = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(keep.sum(1), out=ip[1:])
    vals = (0.01 + 3 * rng.random((n, nnz))).astype(np.float32)
    vals[:600] *= (1 + np.arange(600, dtype=np.float32)[:, None] / 600)
    idx = DeviceIndex.from_csr(ip, cols[keep], vals[keep], V)
    ids, sc = idx.search(q, k)
    allsc = idx.scores(q)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    assert set(ids[0, :min(k, 100)]) <= set(range(600))


@pytest.mark.parametrize("mode", ["0", "1", "bp"], ids=["pairs", "shared-columns", "blocked-postings"])
@pytest.mark.parametrize("store", [nat.VS_F32, nat.VS_F16, nat.VS_NONE], ids=["fp32", "fp16", "binary"])
def test_multi_query_kernel_variants_agree_with_oracle(mode, store):
    """Every variant of the Qt = 8 pass, forced, on ragged rows (0..2000 nnz) and a ragged batch (11 queries)."""
    rng = np.random.default_rng(3)
    n = 3000
    lens = rng.integers(0, 200, size=n)
    lens[::97] = 2000
    lens[5] = 0
    ip = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=ip[1:])
    ix = np.concatenate([np.sort(rng.choice(V, size=l, replace=False)) for l in lens]).astype(np.int32)
    if store == nat.VS_NONE:
        d, dq = None, None
    else:
        d = (0.01 + 3 * rng.random(len(ix))).astype(np.float32)
        if store == nat.VS_F16:
            d = d.astype(np.float16).astype(np.float32)
    q = oracle.synth_queries(1, 11, val_law=synth.VAL_DYADIC if store != nat.VS_F32 else synth.VAL_GRID)
    idx = DeviceIndex.from_csr(ip, ix, d, V, store_dtype=store) if store != nat.VS_NONE else DeviceIndex.from_csr(ip, ix, None, V)
    if mode == "bp":
        idx.set_option("blocked_postings", 1)
    else:
        idx.set_option("blocked_postings", 0)
        idx.set_option("mq_variant", int(mode))
    for k in (1, 64, 300, 1100):
        ids, sc = idx.search(q, k)
        assert idx.info().queries_per_pass == 8
        o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q, k, acc64=True, return_all=True)
        if store == nat.VS_NONE:
            assert (sc == o_sc).all() and (ids == o_ids).all()
        else:
            compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
            compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL)


def test_blocked_postings_auto_policy_and_append():
    """Long-row valued index with >= 16 384 rows: the column-grouped copy is built on the first sparse search (auto policy)
    and must give the CSR scan's results bit for bit; appending rows invalidates and rebuilds it; small binary indexes do not
    build it on their own."""
    n, nnz = 20_000, 300
    ip, ix, d = oracle.synth_csr(5, 0, n + 3000, V, nnz)
    cut = int(ip[n])
    idx = DeviceIndex.reserved(n + 3000, (n + 3000) * ((nnz + 7) // 8), V, nat.VS_F32)
    idx.append_csr(ip[:n + 1], ix[:cut], d[:cut])
    q = oracle.synth_queries(9, 19)
    ids_bp, sc_bp = idx.search(q, 100)
    info = idx.info()
    assert info.last_path == 3 and info.last_fallbacks == 0 and info.aux_bytes > 0 and 0 < info.last_scan_bytes < 3 * info.bytes_per_pass
    idx.set_option("blocked_postings", 0)
    ids_csr, sc_csr = idx.search(q, 100)
    assert idx.info().last_path == 1 and idx.info().aux_bytes == 0
    assert (ids_bp == ids_csr).all() and (sc_bp == sc_csr).all()
    idx.set_option("blocked_postings", -1)
    idx.append_csr(ip[n:] - ip[n], ix[cut:], d[cut:])              # 3000 more rows: the copy is rebuilt on the next search
    ids2, sc2 = idx.search(q, 100)
    assert idx.info().last_path == 3
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, V, q, 100, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids2, sc2, rtol=RTOL)
    compare.compare_topk(o_ids, o_sc, ids2, sc2, rtol=RTOL)
    ipb, ixb, _ = oracle.synth_csr(3, 0, 20_000, V, 300, synth.KIND_BOT)
    bot = DeviceIndex.from_csr(ipb, ixb, None, V)
    bot.search(q, 10)
    assert bot.info().last_path == 1 and bot.info().aux_bytes == 0


@pytest.mark.parametrize("n_cols,qnnz", [(5000, 2000), (257, 200), (31000, 900)])
def test_blocked_postings_other_vocabulary_sizes(n_cols, qnnz):
    """Small / large column counts and dense-ish queries: tiles are planned within the postings kernel's entry capacity."""
    rng = np.random.default_rng(n_cols)
    n, nnz = 3000, min(300, n_cols // 2)
    ix = np.concatenate([np.sort(rng.choice(n_cols, size=nnz, replace=False)) for _ in range(n)]).astype(np.int32)
    ip = np.arange(0, (n + 1) * nnz, nnz, dtype=np.int64)
    d = (0.01 + 3 * rng.random(len(ix))).astype(np.float32)
    q = np.zeros((9, n_cols), dtype=np.float32)
    for b in range(9):
        c = rng.choice(n_cols, size=min(qnnz, n_cols), replace=False)
        q[b, c] = 0.01 + 3 * rng.random(len(c))
    idx = DeviceIndex.from_csr(ip, ix, d, n_cols)
    idx.set_option("blocked_postings", 1)
    ids, sc = idx.search(q, 50)
    o_ids, o_sc, allsc = oracle.csr_search(ip, ix, d, n_cols, q, 50, acc64=True, return_all=True)
    compare.check_topk_valid(allsc, ids, sc, rtol=RTOL)
    compare.compare_topk(o_ids, o_sc, ids, sc, rtol=RTOL)
    assert idx.info().last_path == 3


@pytest.mark.parametrize("kind", [synth.KIND_VDR, synth.KIND_BOT], ids=["sparse", "bag-of-token"])
def test_shard_group_equals_the_unsharded_index(kind):
    """vs_shard_group_*: uneven consecutive row shards (one smaller than k), host and device queries; ids are global and the
    result is the unsharded search bit for bit."""
    import torch
    n = 5000
    nnz, law = (86, synth.VAL_DYADIC) if kind == synth.KIND_BOT else (768, synth.VAL_GRID)
    cuts = [0, 40, 1900, 3100, n]
    full = DeviceIndex.synthetic(7, 0, n, V, nnz, kind, 0, nat.VS_NONE if kind == synth.KIND_BOT else nat.VS_F32)
    shards = [DeviceIndex.synthetic(7, a, b - a, V, nnz, kind, 0, nat.VS_NONE if kind == synth.KIND_BOT else nat.VS_F32) for a, b in zip(cuts[:-1], cuts[1:])]
    group = ShardGroup(shards)
    q = oracle.synth_queries(8, 9, val_law=law)
    want_ids, want_sc = full.search(q, 100)
    ids, sc = group.search(q, 100)
    assert (ids == want_ids).all() and (sc == want_sc).all()
    ids_d, sc_d = group.search(torch.from_numpy(q).cuda(), 100)
    assert (ids_d.cpu().numpy() == want_ids).all() and (sc_d.cpu().numpy() == want_sc).all()
    with pytest.raises(RuntimeError):
        group.search(q, n + 1)
    group.close()
