#!/usr/bin/env python3
"""Per-kernel average time from a rocprofv3 --kernel-trace --stats run:  python tools/prof_kernels.py <dir> [filter]"""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(f)):
    if flt in r["Name"]:
        print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
