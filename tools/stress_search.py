"""Randomised stress of Index.search on the GPU: random index shapes / row-length laws / value stores / batch sizes / k /
column skew, every kernel variant; each result is validated against the scores-only kernel (an independent code path)
and, for binary indexes with dyadic queries, required to be bit-exact.  python tools/stress_search.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vsearch_amd import _native as nat
from vsearch_amd.device_index import DeviceIndex
from oracle import compare

V = 29523


def run(budget=120.0, seed0=0):
    t_end = time.time() + budget
    it = 0
    while time.time() < t_end:
        rng = np.random.default_rng(seed0 * 100003 + it)
        n = int(rng.choice([1, 7, 300, 4000, 30000, 120000]))
        law = rng.integers(0, 4)
        if law == 0:
            lens = np.full(n, int(rng.choice([1, 8, 86, 768])))
        elif law == 1:
            lens = rng.integers(0, int(rng.choice([20, 200, 1500])), size=n)
        elif law == 2:
            lens = np.maximum(1, rng.poisson(86, size=n))
        else:
            lens = rng.integers(0, 40, size=n)
            lens[rng.integers(0, n, size=max(1, n // 50))] = int(rng.choice([1000, 5000, 29523]))
        if n > 30000:
            lens = np.minimum(lens, 64)
        skew = float(rng.choice([0.0, 0.0, 0.7, 1.2]))
        w = 1.0 / np.arange(1, V + 1) ** skew
        cdf = np.cumsum(w / w.sum())
        perm = rng.permutation(V)
        ip = np.zeros(n + 1, dtype=np.int64)
        # distinct columns per row: sample with replacement from the popularity law, unique, pad from a random permutation
        rows = []
        for l in lens:
            l = int(l)
            if l == 0:
                rows.append(np.zeros(0, np.int32)); continue
            c = np.unique(perm[np.searchsorted(cdf, rng.random(min(l * 2, 60000)))])
            if len(c) >= l:
                c = rng.choice(c, size=l, replace=False)
            else:
                extra = np.setdiff1d(rng.permutation(V)[: l + len(c)], c)[: l - len(c)]
                c = np.concatenate([c, extra])
            rows.append(np.sort(c).astype(np.int32))
        lens = np.array([len(r) for r in rows])
        np.cumsum(lens, out=ip[1:])
        ix = np.concatenate(rows) if n else np.zeros(0, np.int32)
        store = int(rng.choice([nat.VS_F32, nat.VS_F16, nat.VS_NONE]))
        dy = store != nat.VS_F32
        d = None
        if store != nat.VS_NONE:
            d = (rng.integers(1, 256, size=len(ix)) / 64).astype(np.float32) if dy else (0.01 + 3 * rng.random(len(ix))).astype(np.float32)
        B = int(rng.choice([1, 2, 8, 9, 33, 70]))
        qn = int(rng.choice([1, 30, 776, 900, 5000]))
        q = np.zeros((B, V), dtype=np.float32)
        for b in range(B):
            c = np.unique(perm[np.searchsorted(cdf, rng.random(qn * 2))])[:qn]
            q[b, c] = (rng.integers(1, 256, size=len(c)) / 64) if dy else (0.01 + 3 * rng.random(len(c)))
        k = int(min(n, rng.choice([1, 5, 100, 257, 600, 2100])))
        idx = DeviceIndex.from_csr(ip, ix, d, V, store_dtype=store) if d is not None else DeviceIndex.from_csr(ip, ix, None, V)
        allsc = idx.scores(q)
        for mode in ("0", "1", "bp", None):
            for qt in (0, 1):
                idx.set_option("blocked_postings", 1 if mode == "bp" else (0 if mode in ("0", "1") else -1))
                idx.set_option("mq_variant", int(mode) if mode in ("0", "1") else -1)
                idx.set_queries_per_pass(qt)
                ids, sc = idx.search(q, k)
                try:
                    # bit-exact only where every partial sum is exact in fp32: binary index x dyadic weights (<= 2^16 hits x 8 bits)
                    ex = store == nat.VS_NONE
                    compare.check_topk_valid(allsc, ids, sc, rtol=1e-4, exact=ex, canonical=ex)
                except AssertionError as e:
                    print(f"FAIL it={it} seed={seed0} n={n} law={law} skew={skew} store={store} B={B} qn={qn} k={k} mode={mode} qt={qt}: {e}", flush=True)
                    return -1
                if qt == 1 and mode is not None:
                    break
        idx.close()
        it += 1
        if it % 10 == 0:
            print(f"{it} cases ok", flush=True)
    print(f"stress ok: {it} random cases", flush=True)
    return it


if __name__ == "__main__":
    sys.exit(0 if run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0) > 0 else 1)
