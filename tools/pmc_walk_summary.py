#!/usr/bin/env python3
"""Sum the counters of tools/pmc_walk.sh for the bench-sized launch of the walk kernel -> profiles/<tag>_utilisation.txt"""
import csv, glob, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_walk"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
kern = sys.argv[3] if len(sys.argv) > 3 else "bp_walk_topk"
shape = sys.argv[4] if len(sys.argv) > 4 else "4 M docs, 1024 queries"
out = {}
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]:
            continue
        per.setdefault((r["Dispatch_Id"], r["Counter_Name"]), 0.0)
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    # the bench-sized launch = the dispatch with the largest counter sum (the same kernel also runs a warm-up on 8 queries and,
    # in its fp64 build, the empty fallback launch)
    tot = {}
    for (d, c), v in per.items():
        tot[d] = tot.get(d, 0.0) + v
    if not tot:
        continue
    big = max(tot, key=tot.get)
    for (d, c), v in per.items():
        if d == big:
            out[c] = v
lines = [f"# rocprofv3 --pmc passes (tools/pmc_walk.sh), largest launch of {kern} ({shape}: the probe of tools/pmc_walk.sh under rocprofv3); sums over all XCDs/SEs"]
for c in sorted(out):
    lines.append(f"{c:44s} {out[c]:24.0f}")
g = out.get
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    lines.append("")
    lines.append(f"wave-cycle split: waiting (s_waitcnt/barrier) {g('SQ_WAIT_ANY', 0)/wc:.3f}, issue stall {g('SQ_WAIT_INST_ANY', 0)/wc:.3f}, issuing {g('SQ_ACTIVE_INST_ANY', 0)/wc:.3f}")
if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_LDS") is not None:
    lines.append(f"SQ_ACTIVE_INST_LDS / SQ_BUSY_CYCLES = {g('SQ_ACTIVE_INST_LDS')/g('SQ_BUSY_CYCLES'):.3f}; VALU {g('SQ_ACTIVE_INST_VALU', 0)/g('SQ_BUSY_CYCLES'):.3f}  (per shader engine: 8 CUs x 4 SIMDs)")
if g("SQ_LDS_IDX_ACTIVE") and g("SQ_BUSY_CYCLES"):
    cu_cycles = g("SQ_BUSY_CYCLES") / 32 * 256                      # SQ_BUSY_CYCLES is summed over 32 shader engines; 256 CUs
    lines.append(f"LDS busy (SQ_LDS_IDX_ACTIVE / CU cycles) = {g('SQ_LDS_IDX_ACTIVE')/cu_cycles:.3f}, of which bank-conflict cycles {g('SQ_LDS_BANK_CONFLICT', 0)/g('SQ_LDS_IDX_ACTIVE'):.3f}")
    if g("SQ_INSTS_LDS_ATOMIC"):
        lines.append(f"VALU instructions per LDS atomic instruction = {g('SQ_INSTS_VALU', 0)/g('SQ_INSTS_LDS_ATOMIC'):.2f}")
if g("TCP_TCC_READ_REQ_sum") and g("SQ_BUSY_CYCLES"):
    cu_cycles = g("SQ_BUSY_CYCLES") / 32 * 256
    lines.append(f"L1->L2 read requests per CU and cycle = {g('TCP_TCC_READ_REQ_sum')/cu_cycles:.3f}; L1 pending-data stall {g('TCP_PENDING_STALL_CYCLES_sum', 0)/cu_cycles:.3f} of the cycles")
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
    lines.append(f"L2 hit rate = {g('TCC_HIT_sum')/(g('TCC_HIT_sum')+g('TCC_MISS_sum')):.3f}")
if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
    lines.append(f"avg TCP->TCC read latency = {g('TCP_TCC_READ_REQ_LATENCY_sum')/g('TCP_TCC_READ_REQ_sum'):.0f} cycles")
if g("SQ_INST_LEVEL_VMEM") and g("SQ_INSTS_VMEM"):
    lines.append(f"avg VMEM instruction latency = {g('SQ_INST_LEVEL_VMEM')/g('SQ_INSTS_VMEM'):.0f} cycles")
# the figures bench.py quotes in its roofline objects: keyed on a hash of the kernel sources, like the traffic passes
import json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rec = {"tag": tag, "shape": shape, "kernel": kern}
if g("TCP_TCC_READ_REQ_sum") and g("SQ_BUSY_CYCLES"):
    rec["l2_requests_per_clk_cu"] = g("TCP_TCC_READ_REQ_sum") / (g("SQ_BUSY_CYCLES") / 32 * 256)
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
    rec["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
    rec["tcp_tcc_read_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
if g("SQ_WAVE_CYCLES"):
    rec["waves_waiting_frac"] = g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")
if g("SQ_LDS_IDX_ACTIVE") and g("SQ_BUSY_CYCLES"):
    rec["lds_busy_frac"] = g("SQ_LDS_IDX_ACTIVE") / (g("SQ_BUSY_CYCLES") / 32 * 256)
if "l2_requests_per_clk_cu" in rec and "l2_hit_rate" in rec:
    # the request ceiling of tools/microbench/l2_requests.hip (profiles/r05_l2_request_rate.txt): 0.40 requests a clock and CU from L2, 0.095 beyond it
    h = rec["l2_hit_rate"]
    rec["request_ceiling_per_clk_cu"] = 1.0 / (h / 0.40 + (1.0 - h) / 0.095)
    rec["request_ceiling_frac"] = rec["l2_requests_per_clk_cu"] / rec["request_ceiling_per_clk_cu"]
try:
    import bench
    rec["source_hash"] = bench.kernel_source_hash()
    js = os.environ.get("VS_PMC_SUMMARY_JSON", os.path.join("profiles", "pmc_summary.json"))
    allrec = json.load(open(js)) if os.path.exists(js) else {}
    allrec.setdefault("utilisation", {})[kern] = rec
    json.dump(allrec, open(js, "w"), indent=1)
except Exception as e:
    print("(pmc_summary.json not updated:", e, ")")
txt = "\n".join(lines) + "\n"
print(txt)
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{tag}_utilisation.txt", "w").write(txt)
