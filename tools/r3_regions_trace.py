"""Mean duration of the region kernels per configuration of tools/r3_regions_time.py (55 calls each, the first 5 dropped),
from the newest rocprofv3 kernel trace under the given directory."""
import csv, glob, os, re, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "region" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
calls, cur = [], None
for r in rows:
    name = re.search(r"k_region_\w+(<[^>]*>)?", r["Kernel_Name"]).group(0)
    if name.startswith("k_region_decode"):
        cur = []; calls.append(cur)
    cur.append((name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for b in range(0, len(calls), 55):
    blk = calls[b:b + 55][5:]
    names = [n for n, _, _, _ in blk[-1]]
    means = [sum(c[i][1] for c in blk) / len(blk) for i in range(len(names))]
    span = sum((c[-1][3] - c[0][2]) / 1e3 for c in blk) / len(blk)
    print(" | ".join(f"{n} {m:.1f}" for n, m in zip(names, means)), f"| first start -> last end {span:.1f} us")
