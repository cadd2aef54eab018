"""World-size-2 test of the row-sharded search protocol on CPU (gloo).  The local search and the
merge are injected from the oracle here (tests may use it; the product binds the HIP entry points),
so what is under test is sharding, global ids, padding, the single all-gather and the merge order.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import V
from vsearch_amd.distributed import ShardedSearcher, pack_candidates, shard_rows, unpack_candidates


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, k, kind, out_dir):
    import oracle
    from vsearch_amd import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        row0, n_local = shard_rows(n_total, world, rank)
        nnz, law = (86, synth.VAL_DYADIC) if kind == synth.KIND_BOT else (768, synth.VAL_GRID)
        ip, ix, d = oracle.synth_csr(11, row0, n_local, V, nnz, kind)
        data = None if kind == synth.KIND_BOT else d

        def local_search(q, kk, off):
            ids, sc = oracle.csr_search(ip, ix, data, V, q.numpy(), kk)
            return torch.from_numpy(ids + off), torch.from_numpy(sc)

        def merge(ci, cs, kk):
            ids, sc = oracle.merge_topk(ci.numpy(), cs.numpy(), kk)
            return torch.from_numpy(ids), torch.from_numpy(sc)

        searcher = ShardedSearcher(local_search, merge, n_local, row0, n_total)
        q = torch.from_numpy(oracle.synth_queries(12, 5, val_law=law))
        ids, sc = searcher.search(q, k)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=ids.numpy(), sc=sc.numpy())
        if rank == 0:
            with pytest.raises(RuntimeError):
                searcher.search(q, n_total + 1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,k,kind", [(1201, 100, 0), (150, 100, 0), (3000, 100, 1)])
def test_sharded_search_equals_unsharded(tmp_path, n_total, k, kind):
    import oracle
    from vsearch_amd import synth
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, k, kind, str(tmp_path)), nprocs=world, join=True)
    nnz, law = (86, synth.VAL_DYADIC) if kind == synth.KIND_BOT else (768, synth.VAL_GRID)
    ip, ix, d = oracle.synth_csr(11, 0, n_total, V, nnz, kind)
    q = oracle.synth_queries(12, 5, val_law=law)
    want_ids, want_sc = oracle.csr_search(ip, ix, None if kind == synth.KIND_BOT else d, V, q, k)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        assert (got["ids"] == want_ids).all(), f"rank {r}: ids differ from the unsharded search"
        assert (got["sc"] == want_sc).all()


def test_shard_rows_partition():
    for n, w in [(21015324, 8), (10, 3), (5, 8), (0, 2)]:
        spans = [shard_rows(n, w, r) for r in range(w)]
        assert sum(c for _, c in spans) == n
        pos = 0
        for r0, c in spans:
            assert r0 == pos or c == 0
            pos += c


def test_pack_unpack_roundtrip():
    ids = torch.tensor([[0, 1, 21015323, (1 << 32) - 1]], dtype=torch.int64)
    sc = torch.tensor([[0.0, -1.5, 3.25e8, float("-inf")]])
    i2, s2 = unpack_candidates(pack_candidates(ids, sc))
    assert (i2 == ids).all() and (s2 == sc).all()
