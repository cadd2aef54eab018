"""Strong-scaling forecast on ONE GPU: the per-rank shard of the 21 015 324-doc index at 1, 2, 4, 8 ranks is searched with the full
1024-query batch; ranks run in parallel, so the job's rate is 1024 / (slowest shard's time + exchange).  The exchange (one
all-gather of B x k packed candidates per rank + merge) is measured on a 1-rank nccl group where available.
python tools/probe_scaling.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, merge_topk
from vsearch_amd.distributed import shard_rows
import oracle

N, B, K = 21_015_324, 1024, 100
q = torch.from_numpy(oracle.synth_queries(1, B)).cuda()
base = None
for g in (1, 2, 4, 8):
    row0, n_local = shard_rows(N, g, g - 1)                     # the last rank's shard (the largest differs by at most one row)
    idx = DeviceIndex.synthetic(0, row0, n_local, 29523, 768, 0, 0, nat.VS_F32)
    idx.search(q[:8], K)
    idx.search(q, K)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3):
        ids, sc = idx.search(q, K, id_offset=row0)
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    # merge of g ranks' candidates (what every rank does after the all-gather)
    ci = ids.repeat(1, g).contiguous(); cs = sc.repeat(1, g).contiguous()
    merge_topk(ci, cs, K); torch.cuda.synchronize(); t = time.time(); merge_topk(ci, cs, K); torch.cuda.synchronize(); mt = time.time() - t
    rate = B / (dt + (mt if g > 1 else 0))
    base = base or rate
    print(f"ranks={g}: shard {n_local} docs  search {dt*1e3:.1f} ms  merge {mt*1e3:.2f} ms  -> {rate:.0f} q/s  ({rate/base/g*100:.0f} % of linear)", flush=True)
    idx.close()
