"""Skewed column popularity (SURVEY §8(d) C3 secondary run): Zipf(s) column law for documents AND queries.
python tools/probe_skew.py [N] [B] [s ...]   -- builds the CSR with torch on the GPU, searches, validates against the
scores-only kernel, prints q/s for the Qt = 8 pass and the Qt = 1 pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex, Profile
from oracle import compare

V, A, K = 29523, 768, 100


def zipf_rows(n, nnz, s, gen, chunk=4096):
    w = (1.0 / torch.arange(1, V + 1, device="cuda", dtype=torch.float64) ** s).float()
    perm = torch.randperm(V, device="cuda", generator=gen)            # popularity rank -> column id
    cols = []
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        c = torch.multinomial(w.expand(m, V), nnz, replacement=False, generator=gen)
        cols.append(perm[c].sort(dim=1).values)
    return torch.cat(cols)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    ss = [float(x) for x in sys.argv[3:]] or [0.0, 0.5, 1.0]
    nat.require_device()
    for s in ss:
        gen = torch.Generator(device="cuda").manual_seed(0)
        w = (1.0 / torch.arange(1, V + 1, device="cuda", dtype=torch.float64) ** s).float()
        perm = torch.randperm(V, device="cuda", generator=gen)
        cols = []
        for a in range(0, N, 4096):
            m = min(4096, N - a)
            cols.append(perm[torch.multinomial(w.expand(m, V), A, replacement=False, generator=gen)].sort(dim=1).values)
        cols = torch.cat(cols)
        vals = 0.01 + 3 * torch.rand(cols.shape, device="cuda", generator=gen)
        crow = torch.arange(0, (N + 1) * A, A, device="cuda", dtype=torch.int64)
        idx = DeviceIndex.from_csr(crow, cols.reshape(-1), vals.reshape(-1), V)
        qc = perm[torch.multinomial(w.expand(B, V), A + 8, replacement=False, generator=gen)]
        q = torch.zeros((B, V), device="cuda")
        q.scatter_(1, qc, 0.01 + 3 * torch.rand(qc.shape, device="cuda", generator=gen))
        del cols, vals
        for qt, shared in ((0, "0"), (0, "1"), (0, None), (0, "bp"), (1, None)):
            idx.set_option("blocked_postings", 1 if shared == "bp" else 0)
            idx.set_option("mq_variant", int(shared) if shared in ("0", "1") else -1)
            idx.set_queries_per_pass(qt)
            idx.search(q, K)
            torch.cuda.synchronize(); t = time.time()
            ids, sc = idx.search(q, K)
            torch.cuda.synchronize(); dt = time.time() - t
            allsc = idx.scores(q[:4])
            allsc = allsc.cpu().numpy() if hasattr(allsc, "cpu") else allsc
            compare.check_topk_valid(allsc, ids[:4].cpu().numpy(), sc[:4].cpu().numpy(), rtol=1e-4)
            print(f"zipf s={s}: N={N} B={B} qt_pref={qt} mode={shared} used_qt={idx.info().queries_per_pass}  {dt*1e3:.1f} ms  {B/dt:.0f} q/s  (top-k valid)", flush=True)
        idx.close()


if __name__ == "__main__":
    main()
