"""Rank process for tests/test_launcher_cpu.py: started by vsearch_amd.launch.spawn_ranks exactly as bench.py's ranks are
(RANK / WORLD_SIZE / MASTER_* from the environment), gloo instead of RCCL, oracle-injected local search."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    mode, out_dir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "fail" and rank == 1:
        sys.exit(3)
    import numpy as np
    import torch
    import torch.distributed as dist
    import oracle
    from vsearch_amd.distributed import ShardedSearcher, shard_rows
    dist.init_process_group("gloo")                    # env:// rendezvous, like bench.py
    try:
        if mode == "fail":
            dist.barrier()                             # rank 1 never arrives: the launcher must terminate us
            return
        V, n_total, k = 29523, 1501, 100
        row0, n_local = shard_rows(n_total, world, rank)
        ip, ix, d = oracle.synth_csr(21, row0, n_local, V, 768, 0)

        def local_search(q, kk, off):
            ids, sc = oracle.csr_search(ip, ix, d, V, q.numpy(), kk)
            return torch.from_numpy(ids + off), torch.from_numpy(sc)

        def merge(ci, cs, kk):
            ids, sc = oracle.merge_topk(ci.numpy(), cs.numpy(), kk)
            return torch.from_numpy(ids), torch.from_numpy(sc)

        q = torch.from_numpy(oracle.synth_queries(22, 4))
        ids, sc = ShardedSearcher(local_search, merge, n_local, row0, n_total).search(q, k)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=ids.numpy(), sc=sc.numpy(), world=dist.get_world_size())
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
