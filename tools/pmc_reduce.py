#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs to per-kernel averages (our kernels only).

    python tools/pmc_reduce.py <dir with *counter_collection.csv ...> > summary.json

Sums a counter over a dispatch's rows (rocprofv3 emits one row per dispatch and counter
instance), then averages over the dispatches of each kernel.  Kernel names are shortened to the
function name."""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+)", name)
    return m.group(1) + ("<" + name.split("<", 1)[1].split(">", 1)[0] + ">" if m and "<" in name.split("(")[0] else "") if m else None


def main():
    per = defaultdict(lambda: defaultdict(dict))           # kernel -> counter -> dispatch -> value
    dur = defaultdict(dict)
    for root in sys.argv[1:]:
        for f in Path(root).rglob("*counter_collection.csv"):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if not k:
                        continue
                    d = (str(f), row["Dispatch_Id"])
                    c = row["Counter_Name"]
                    per[k][c][d] = per[k][c].get(d, 0.0) + float(row["Counter_Value"])
                    if row.get("End_Timestamp") and row.get("Start_Timestamp"):
                        dur[k][d] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    out = {}
    for k, cs in sorted(per.items()):
        out[k] = {c: {"mean": sum(v.values()) / len(v), "launches": len(v)} for c, v in sorted(cs.items())}
        if dur[k]:
            out[k]["_us_in_profiled_pass"] = sum(dur[k].values()) / len(dur[k])
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
