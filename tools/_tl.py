import csv, re, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stats_fill_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); pe = None
for r in rows[a - 5:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    m = re.search(r"(\w+)(<[^(]*)?\(", r["Kernel_Name"]); name = m.group(1) if m else r["Kernel_Name"][:40]
    print(f"{name:30s} q{r['Queue_Id']} start {s/1000:9.1f} dur {(e-s)/1000:7.1f} gap {'' if pe is None else round((s-pe)/1000,1)}")
    pe = e
