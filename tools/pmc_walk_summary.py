#!/usr/bin/env python3
"""Sum the counters of tools/pmc_walk.sh for the bench-sized launch of the walk kernel -> profiles/<tag>_utilisation.txt"""
import csv, glob, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_walk"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
kern = sys.argv[3] if len(sys.argv) > 3 else "bp_scan_topk"
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
lines = [f"# rocprofv3 --pmc passes (tools/pmc_walk.sh), largest (bench-sized: 21 M docs, 1024 queries) launch of {kern}; sums over all XCDs/SEs"]
for c in sorted(out):
    lines.append(f"{c:44s} {out[c]:24.0f}")
g = out.get
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    lines.append("")
    lines.append(f"wave-cycle split: waiting (s_waitcnt/barrier) {g('SQ_WAIT_ANY', 0)/wc:.3f}, issue stall {g('SQ_WAIT_INST_ANY', 0)/wc:.3f}, issuing {g('SQ_ACTIVE_INST_ANY', 0)/wc:.3f}")
if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_LDS") is not None:
    lines.append(f"SQ_ACTIVE_INST_LDS / SQ_BUSY_CYCLES = {g('SQ_ACTIVE_INST_LDS')/g('SQ_BUSY_CYCLES'):.3f}; VALU {g('SQ_ACTIVE_INST_VALU', 0)/g('SQ_BUSY_CYCLES'):.3f}; VMEM {g('SQ_ACTIVE_INST_VMEM', 0)/g('SQ_BUSY_CYCLES'):.3f}")
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
    lines.append(f"L2 hit rate = {g('TCC_HIT_sum')/(g('TCC_HIT_sum')+g('TCC_MISS_sum')):.3f}")
if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
    lines.append(f"avg TCP->TCC read latency = {g('TCP_TCC_READ_REQ_LATENCY_sum')/g('TCP_TCC_READ_REQ_sum'):.0f} cycles")
if g("SQ_INST_LEVEL_VMEM") and g("SQ_INSTS_VMEM"):
    lines.append(f"avg VMEM instruction latency = {g('SQ_INST_LEVEL_VMEM')/g('SQ_INSTS_VMEM'):.0f} cycles")
txt = "\n".join(lines) + "\n"
print(txt)
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{tag}_utilisation.txt", "w").write(txt)
