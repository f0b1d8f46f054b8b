#!/usr/bin/env python
"""Per kernel name: launches and the median duration of its LONGEST half of launches (large problem sizes
dominate; tiny calls of the same kernel are ignored).   python tools/dbg/ktrace_top.py <kernel_trace.csv> [filter]"""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
flt = sys.argv[2] if len(sys.argv) > 2 else "fl::"
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if flt in k:
        v.sort()
        top = v[len(v) // 2:]
        print("%-90s n=%4d  upper-half median %8.1f us" % (k[:90], len(v), top[len(top) // 2]))
