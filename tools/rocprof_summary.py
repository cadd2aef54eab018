#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (gpurun_out/<dir>/*_results.db) into the small text / JSON
summaries committed under profiles/.

    python tools/rocprof_summary.py --stats gpurun_out/prof_stats/r01_results.db \
        --fetch gpurun_out/prof_fetch/r01_results.db --write gpurun_out/prof_write/r01_results.db \
        --tag r01 --note "bench.py --steps 2 --warmup 1"

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB,
collected in separate --pmc passes; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
(16 B/lane) coalesced streaming read, so the read side is doubled.
"""
import argparse
import json
import os
import sqlite3


def source_hash():
    """bench.py's kernel_source_hash(): the profile is quoted only for the kernel sources it was taken on."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.kernel_source_hash()


def top_kernels(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    per = con.execute("select name, (end - start) from kernels order by start").fetchall()
    return rows, per


def counters(db, counter):
    con = sqlite3.connect(db)
    return con.execute("select kernel_name, grid_size, sum(value), count(*) from counters_collection where counter_name = ? "
                       "group by kernel_name, dispatch_id order by dispatch_id", (counter,)).fetchall()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--note", default="")
    ap.add_argument("--queries", type=int, default=0, help="queries in the bench-sized scan launch: with --fetch and --write, updates "
                    "profiles/pmc_summary.json (read by bench.py for roofline.traffic)")
    ap.add_argument("--docs", type=int, default=21015324)
    ap.add_argument("--store", default="fp32", help="bench.py --store of the profiled command")
    ap.add_argument("--scan", default="auto", help="bench.py --scan of the profiled command")
    ap.add_argument("--columns", default="uniform", help="bench.py --columns of the profiled command")
    ap.add_argument("--leg", default="", help="a secondary leg of bench.py (tools/leg_pmc.py under rocprofv3): key in bench.SECONDARY_LEGS, e.g. C3_1m_sparse; "
                    "with --fetch and --write, updates profiles/pmc_summary.json -> legs[key]")
    ap.add_argument("--searches", type=int, default=2, help="--leg: full-batch searches in the profiled command (tools/leg_pmc.py: 1 warm-up + N)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    summary = {"tag": a.tag, "note": a.note}
    if a.stats:
        rows, per = top_kernels(a.stats)
        with open(os.path.join(a.out, f"{a.tag}_kernel_stats.txt"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats   ({a.note})\n# durations in milliseconds\n")
            f.write(f"{'kernel':100s} {'calls':>6s} {'total_ms':>14s} {'avg_ms':>14s} {'pct':>7s}\n")
            for name, calls, tot, avg, pct in rows:            # top_kernels reports microseconds
                f.write(f"{name[:100]:100s} {calls:6d} {tot / 1e3:14.3f} {avg / 1e3:14.3f} {pct:7.3f}\n")
            big = {}
            for name, dur in per:
                if dur >= 50e6:                                # bench-sized launches only (>= 50 ms)
                    big.setdefault(name, []).append(dur / 1e6)
            f.write("\n# bench-sized launches only (>= 50 ms; excludes the 20 k-doc parity check that reuses the kernel)\n")
            for name, ds in big.items():
                f.write(f"{name[:100]:100s} {len(ds):6d} launches, avg {sum(ds) / len(ds):12.3f} ms\n")
            summary["bench_sized_launches"] = {n: {"launches": len(d), "avg_ms": sum(d) / len(d)} for n, d in big.items()}
            f.write("\n# per dispatch (launch order), ms\n")
            for name, dur in per:                              # kernels view: end - start in nanoseconds
                f.write(f"{name[:100]:100s} {dur / 1e6:14.3f}\n")
        summary["kernel_stats"] = [{"kernel": r[0], "calls": r[1], "total_ms": r[2] / 1e3, "avg_ms": r[3] / 1e3, "pct": r[4]} for r in rows]
        summary["dispatches"] = [{"kernel": n, "ms": d / 1e6} for n, d in per]
    for key, db, counter in (("fetch", a.fetch, "FETCH_SIZE"), ("write", a.write, "WRITE_SIZE")):
        if db:
            rows = counters(db, counter)
            summary[counter] = [{"kernel": r[0], "grid": r[1], "KiB": r[2]} for r in rows]
            with open(os.path.join(a.out, f"{a.tag}_{counter.lower()}.txt"), "w") as f:
                f.write(f"# rocprofv3 --pmc {counter}   ({a.note})\n# value = sum over XCDs/instances, KiB; per dispatch in launch order\n")
                for name, grid, val, n in rows:
                    f.write(f"{name[:100]:100s} grid={grid:<10d} {val:18.1f}\n")
    if a.leg and "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        # a secondary leg: HBM bytes of ONE full-batch search = sum over the bench-sized dispatches (>= 1/5 of the largest: the 8-query
        # warm-up search is excluded) of the leg's scan kernel (+ its head pre-pass), divided by the full-batch searches profiled
        pmc_path = os.path.join(a.out, "pmc_summary.json")
        pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
        scan_kernels = ("bp_quad_topk", "bp_walk_topk", "bp_bin_topk", "bp_bq_topk", "csr_scan_topk_mq", "head_gemm")
        def total(rows):
            vals = [r["KiB"] for r in rows if any(k in r["kernel"] for k in scan_kernels)]
            big = [v for v in vals if v >= 0.2 * max(vals)] if vals else []
            return sum(big) / max(1, a.searches), len(big)
        fetch, nf = total(summary["FETCH_SIZE"])
        write, _ = total(summary["WRITE_SIZE"])
        names = sorted({r["kernel"].split("<")[0].split("(")[0].replace("void vs::", "") for r in summary["FETCH_SIZE"] if any(k in r["kernel"] for k in scan_kernels)})
        if fetch > 0:
            pmc.setdefault("legs", {})[a.leg] = {
                "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "fetch_KiB": fetch, "write_KiB": write, "queries_per_launch": a.queries or 1024,
                "kernels": names, "bench_sized_dispatches": nf, "searches": a.searches, "tag": a.tag, "source_hash": source_hash(),
                "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 summed over the scan dispatches of one full-batch search",
                "source": f"profiles/{a.tag}_fetch_size.txt, profiles/{a.tag}_write_size.txt (rocprofv3 --pmc, separate passes, {a.note})"}
            with open(pmc_path, "w") as f:
                json.dump(pmc, f, indent=1)
    elif a.queries > 0 and "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        # bench-sized launch (largest) of each scan kernel present -> profiles/pmc_summary.json, read by bench.py for roofline.traffic
        pmc_path = os.path.join(a.out, "pmc_summary.json")
        pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
        scan_kernels = ("bp_quad_topk", "bp_walk_topk", "bp_bin_topk", "bp_bq_topk", "csr_scan_topk_mq")
        pmc = {k: v for k, v in pmc.items() if k in scan_kernels or k == "legs"}
        for key in scan_kernels:
            scan = lambda rows: max((r["KiB"] for r in rows if key in r["kernel"]), default=0.0)
            fetch, write = scan(summary["FETCH_SIZE"]), scan(summary["WRITE_SIZE"])
            if fetch <= 0:
                continue
            pmc[key] = {"hbm_bytes_per_launch": (2 * fetch + write) * 1024, "fetch_KiB": fetch, "write_KiB": write,
                        "queries_per_launch": a.queries, "docs": a.docs, "store": a.store, "scan": a.scan, "columns": a.columns, "tag": a.tag,
                        "source_hash": source_hash(),
                        "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 1/2 of a 16 B/lane coalesced stream)",
                        "source": f"profiles/{a.tag}_fetch_size.txt, profiles/{a.tag}_write_size.txt (rocprofv3 --pmc, separate passes, {a.note})"}
        with open(pmc_path, "w") as f:
            json.dump(pmc, f, indent=1)
    with open(os.path.join(a.out, f"{a.tag}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k in ("tag", "note")}))


if __name__ == "__main__":
    main()
