#!/usr/bin/env python3
"""Print per-kernel PMC counter sums from a rocprofv3 rocpd database: python tools/pmc_dump.py <db> [kernel-substring]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = con.execute("select kernel_name, dispatch_id, counter_name, sum(value), max(end-start) from counters_collection group by kernel_name, dispatch_id, counter_name order by dispatch_id").fetchall()
for name, did, cname, val, dur in rows:
    if pat in name:
        print(f"{did:4d} {name[:60]:60s} {cname:28s} {val:18.0f}  dur_ns={dur}")
