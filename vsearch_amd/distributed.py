"""Row-sharded search across the GPUs of one node (new in this build).

The reference searches one index on one device; its only sharding precedent is the per-shard index
*build* (``--num_shard/--shard_id`` -> one ``.npz`` per shard, re-joined by ``vstack`` at
/root/reference/src/ir/retriever/index.py:172-175).  Here documents (CSR rows) are split into
contiguous row ranges, one per rank (one process per GPU, ``torch.distributed`` -- backend "nccl" is
RCCL over xGMI); every rank scores the whole query batch against its rows, then ONE all-gather
moves B*k packed (score, global id) pairs per rank (<= 0.8 MB at B = 1024, k = 100: latency-bound on
the fully connected 7-link topology, no ring, no all-reduce) and every rank selects the final
top-k.  With the canonical order (score desc, id asc) the result is identical to searching the
unsharded index.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

_PAD_ID = (1 << 32) - 1        # sentinel for shards with fewer than k rows: loses every comparison


def shard_rows(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous equal row ranges: -> (row0, n_rows) of `rank`."""
    per = -(-n_total // world)
    row0 = min(n_total, rank * per)
    return row0, max(0, min(n_total, row0 + per) - row0)


def pack_candidates(ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
    """(int64 ids < 2^32, fp32 scores) -> one int64 per candidate (score bits high, id low)."""
    hi = scores.contiguous().view(torch.int32).to(torch.int64) << 32
    return hi | (ids & 0xFFFFFFFF)


def unpack_candidates(packed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    ids = packed & 0xFFFFFFFF
    scores = (packed >> 32).to(torch.int32).view(torch.float32)
    return ids, scores


class ShardedSearcher:
    """One rank's view of a row-sharded index.

    local_search(q, k, id_offset) -> (ids, scores): this rank's top-k with GLOBAL ids (the HIP
        ``DeviceIndex.search`` bound to the local shard);
    merge(cand_ids, cand_scores, k) -> (ids, scores): canonical top-k of [B, world*k] candidates
        (``device_index.merge_topk``).
    Both are injectable so the protocol can be exercised on CPU (gloo) by the tests.
    """

    def __init__(self, local_search: Callable, merge: Callable, n_local: int, row0: int, n_total: int,
                 group: Optional[torch.distributed.ProcessGroup] = None):
        self.local_search, self.merge = local_search, merge
        self.n_local, self.row0, self.n_total, self.group = int(n_local), int(row0), int(n_total), group
        if self.n_total >= _PAD_ID:
            # candidates travel as (score bits << 32 | global id): ids are 32-bit on the wire and 2^32 - 1 is the pad sentinel
            raise ValueError(f"row-sharded search addresses at most {_PAD_ID - 1} documents (n_total = {self.n_total})")
        self.force_exchange = False     # tests: run the all-gather + merge even in a 1-rank group
        self._events = None             # enable_timing(): per search (start, after local search, after exchange, after merge) CUDA events

    def enable_timing(self, on: bool = True):
        """Record CUDA events around the three phases of every search (no synchronisation inside the search): read_timing()
        sums them afterwards -- the per-rank breakdown bench.py prints for N > 1."""
        self._events = [] if on else None

    def read_timing(self):
        """-> {"local_ms", "exchange_ms", "merge_ms", "searches"} summed over the searches since enable_timing(); synchronises."""
        out = {"local_ms": 0.0, "exchange_ms": 0.0, "merge_ms": 0.0, "searches": 0}
        if not self._events:
            return out
        torch.cuda.synchronize()
        for e0, e1, e2, e3 in self._events:
            out["local_ms"] += e0.elapsed_time(e1)
            out["exchange_ms"] += e1.elapsed_time(e2)
            out["merge_ms"] += e2.elapsed_time(e3)
            out["searches"] += 1
        return out

    def _mark(self, q):
        if self._events is None or q.device.type != "cuda":
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    @classmethod
    def from_device_index(cls, index, row0: int, n_total: int, group=None):
        from .device_index import merge_topk
        info = index.info()
        return cls(lambda q, k, off: index.search(q, k, id_offset=off), lambda i, s, k: merge_topk(i, s, k, device=info.device),
                   info.n_rows, row0, n_total, group)

    def search(self, q: torch.Tensor, k: int):
        import torch.distributed as dist
        if k > self.n_total:
            raise RuntimeError(f"selected index k out of range (k = {k} > {self.n_total} rows)")
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        k_local = min(k, self.n_local)
        B = q.shape[0]
        e0 = self._mark(q)
        if k_local > 0:
            ids, scores = self.local_search(q, k_local, self.row0)
        else:
            ids = torch.empty((B, 0), dtype=torch.int64, device=q.device)
            scores = torch.empty((B, 0), dtype=torch.float32, device=q.device)
        e1 = self._mark(q)
        if world == 1 and not self.force_exchange:
            if e0 is not None:
                self._events.append((e0, e1, e1, e1))
            return ids, scores
        if k_local < k:                      # pad so that every rank contributes exactly k slots
            pad = k - k_local
            ids = torch.cat([ids, torch.full((B, pad), _PAD_ID, dtype=torch.int64, device=ids.device)], 1)
            scores = torch.cat([scores, torch.full((B, pad), float("-inf"), dtype=torch.float32, device=scores.device)], 1)
        packed = pack_candidates(ids, scores).contiguous()
        dev = packed.device
        if dev.type == "cuda" and dist.is_initialized() and dist.get_backend(self.group) == "gloo":
            packed = packed.cpu()            # test rigs only (ranks sharing one GPU): gloo moves host memory
        gathered = torch.empty((world * B, k), dtype=torch.int64, device=packed.device)   # rank-major concatenation
        dist.all_gather_into_tensor(gathered, packed, group=self.group)        # the one exchange step
        gathered = gathered.to(dev)
        e2 = self._mark(q)
        cand_ids, cand_scores = unpack_candidates(gathered.view(world, B, k).permute(1, 0, 2).reshape(B, world * k).contiguous())
        out = self.merge(cand_ids.contiguous(), cand_scores.contiguous(), k)
        if e0 is not None:
            self._events.append((e0, e1, e2, self._mark(q)))
        return out
