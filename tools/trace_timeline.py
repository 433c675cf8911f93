#!/usr/bin/env python3
"""Timeline of the last complete call in a rocprofv3 --kernel-trace CSV: start, duration and gap of every kernel
between two launches of the depth smoothing kernel.  usage: trace_timeline.py <kernel_trace.csv> [calls back]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "smooth_depth" in r["Kernel_Name"]]
a, b = idx[-back], idx[-back + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    m = re.search(r"(\w+)(<[^(]*)?\(", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    gap = "" if prev_end is None else f"{(s - prev_end) / 1000:6.1f}"
    print(f"{name:28s} queue {r['Queue_Id']} start {s / 1000:8.1f} us  duration {(e - s) / 1000:6.1f}  gap {gap}")
    prev_end = e
