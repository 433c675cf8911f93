#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].replace("void ", "").replace("fs::(anonymous namespace)::", "").split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
for name, ctrs in sorted(acc.items()):
    n = max(len(v) for v in ctrs.values())
    lines.append(f"{name}  (dispatches {n})")
    for c, v in sorted(ctrs.items()):
        lines.append(f"    {c:28s} mean {sum(v)/len(v):16.1f}")
txt = "\n".join(lines)
open(os.path.join(root, "summary.txt"), "w").write(txt + "\n")
print(txt)
