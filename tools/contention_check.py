"""N processes sharing ONE GPU, each searching its own synthetic shard over and over: every filter + refine result against the CSR scan of
the same index.  python tools/contention_check.py [processes] [searches] [plain|bot|zipf|dense|mask]
(dense: the MFMA search of a dense index against torch; mask: the encoder's mask stage and the fused mask -> CSR kernel -- whose workgroups
wait for each other's totals in ticket order: under contention they are not all resident at once -- against torch.)

Why: with four or more processes other kernels' waves share the CUs and stretch the timing windows inside a workgroup; round 5 had a
version of the quad walk that passed every single-process test and lost whole blocks of candidates in 10 - 25 % of the searches here
(root cause, round 6: a per-thread cut decision in the walks' epilogue -- docs/EXPERIMENTS.md, profiles/r06_prune_decision_race.txt).
tests/test_gpu_search.py runs it."""
import os, subprocess, sys
import numpy as np
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, repo)

KINDS = {"plain": (20_000, 768, 0, "VS_F32"), "bot": (60_000, 86, 1, "VS_NONE"), "zipf": (40_000, 768, 2, "VS_F32")}


def child_dense(rank, reps):
    import torch
    from vsearch_amd.device_index import DeviceIndex
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    V, n, b = 4096, 30_000, 64
    mat = torch.zeros((n, V), device="cuda")
    c = torch.rand((n, V), device="cuda", generator=g).topk(96, dim=1).indices
    mat.scatter_(1, c, 0.01 + 3 * torch.rand(c.shape, device="cuda", generator=g))
    q = torch.zeros((b, V), device="cuda")
    qc = torch.rand((b, V), device="cuda", generator=g).topk(64, dim=1).indices
    q.scatter_(1, qc, 0.01 + 3 * torch.rand(qc.shape, device="cuda", generator=g))
    idx = DeviceIndex.from_dense(mat)
    want = (q.double() @ mat.double().T).topk(100, dim=1)
    nbad = 0
    for i in range(reps):
        ids, sc = idx.search(q, 100)
        # (scores to 1e-4 against fp64; ids where the fp64 scores are not tied to that tolerance)
        ok = torch.allclose(sc.double(), want.values, rtol=1e-4, atol=0) and bool(((ids == want.indices) | (torch.abs(sc.double() - want.values) <= 1e-4 * want.values)).all())
        nbad += 0 if ok else 1
    print(f"rank {rank} dense: {nbad} bad searches of {reps} (path dense, walk -)", flush=True)
    return 1 if nbad else 0


def child_mask(rank, reps):
    import torch
    from vsearch_amd.ir.utils import sparse as sp
    g = torch.Generator(device="cuda").manual_seed(200 + rank)
    B, V, K, L, VOC, SHIFT = 700, 29523, 768, 64, 30522, 999
    emb = torch.rand((B, V), device="cuda", generator=g) * 3
    emb[5] = 0.0
    tok = torch.randint(SHIFT, VOC, (B, L), device="cuda", generator=g)
    mask = torch.zeros_like(emb, dtype=torch.bool)
    mask.scatter_(1, emb.topk(K, dim=1).indices, True)
    mask.scatter_(1, tok - SHIFT, True)
    dense = emb * mask
    want = dense.to_sparse_csr()
    nbad = 0
    for i in range(reps):
        rp, ci, va = sp.embed_mask_to_csr(emb, tok, VOC, SHIFT, K, True)
        ok = bool((rp == want.crow_indices()).all()) and ci.numel() == want.col_indices().numel() and bool((ci == want.col_indices()).all()) and bool((va == want.values()).all())
        e2 = emb.clone()
        sp.apply_embed_mask_(e2, tok, VOC, SHIFT, K, True)
        ok = ok and bool((e2 == dense).all())
        nbad += 0 if ok else 1
    print(f"rank {rank} mask: {nbad} bad searches of {reps} (path mask, walk -)", flush=True)
    return 1 if nbad else 0


def child(rank, reps, kind):
    if kind == "dense": return child_dense(rank, reps)
    if kind == "mask": return child_mask(rank, reps)
    import bench, torch
    from collections import Counter
    from vsearch_amd import _native as nat
    from vsearch_amd.device_index import DeviceIndex
    n, nnz, k, store = KINDS[kind]
    idx = DeviceIndex.synthetic(bench.INDEX_SEED, rank * n, n, 29523, nnz, k, 0, getattr(nat, store))
    qb = bench.make_query_batches(2, 32, torch.device("cuda", 0), kind=k)
    idx.set_option("blocked_postings", 0)
    ref = []
    for q in qb:
        ids, sc = idx.search(q, 100)
        ref.append((ids.cpu().numpy().copy(), sc.cpu().numpy().copy()))
    idx.set_option("blocked_postings", 1)
    for kv in filter(None, os.environ.get("VS_CHECK_OPTS", "").split(",")):
        name, val = kv.split("=")
        idx.set_option(name, int(val))
    nbad = 0
    for i in range(reps):
        ids, sc = idx.search(qb[i & 1], 100)
        got, want = (ids.cpu().numpy(), sc.cpu().numpy()), ref[i & 1]
        bad = np.argwhere((got[0] != want[0]) | (got[1] != want[1]))
        if len(bad):
            nbad += 1
            if nbad <= 3:
                miss = sorted(set(want[0].ravel().tolist()) - set(got[0].ravel().tolist()))
                print(f"rank {rank} search {i}: {len(bad)} mismatches; missing documents by 2048-row block: {sorted(Counter(m // 2048 for m in miss).items())}", flush=True)
    inf = idx.info()
    print(f"rank {rank} {kind}: {nbad} bad searches of {reps} (path {inf.last_path}, walk {inf.postings_walk})", flush=True)
    return 1 if nbad else 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        raise SystemExit(child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    kind = sys.argv[3] if len(sys.argv) > 3 else "plain"
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(r), str(reps), kind]) for r in range(n)]
    raise SystemExit(max(p.wait() for p in ps))
