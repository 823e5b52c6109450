#!/usr/bin/env python3
"""Reduce the passes of tools/ta_roof.sh to profiles/ta_roof.json (what bench.py's roofline_ta reads).

peak_lines_per_s: cache-line tag accesses (TCP_TOTAL_CACHE_ACCESSES) per second the chip sustains when every CU streams whole
lines (csrc/lab/ta_lab map 3: 1 KB contiguous per wave-instruction), lines from the PMC pass, time from the un-profiled wall-clock
run of the same binary.  Per kernel of the metric workload: lines per launch from the PMC pass of bench.py."""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path


def pmc(root):
    per = defaultdict(lambda: defaultdict(dict))
    dur = defaultdict(dict)
    for f in Path(root).rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                m = re.search(r"(k_[A-Za-z0-9_]+(?:<[^>]*>)?)", row["Kernel_Name"])
                if not m:
                    continue
                k, d = m.group(1), row["Dispatch_Id"]
                per[k][row["Counter_Name"]][d] = per[k][row["Counter_Name"]].get(d, 0.0) + float(row["Counter_Value"])
                if row.get("End_Timestamp") and row.get("Start_Timestamp"):
                    dur[k][d] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    return per, dur


def main():
    out = Path(sys.argv[1])
    lab, lab_dur = {}, {}
    for d in ("lab_pmc1", "lab_pmc2"):
        p, du = pmc(out / d)
        for k, cs in p.items():
            lab.setdefault(k, {}).update({c: v for c, v in cs.items()})
            lab_dur[k] = du[k]
    # ta_lab launches every map as (warm 10 iterations, timed 400 iterations) x 3: the long dispatches are the timed ones
    wall = {}
    for line in (out / "ta_lab_wall.txt").read_text().splitlines():
        m = re.match(r"map (\d).*?([\d.]+) us\s+=", line)
        if m:
            wall.setdefault(int(m.group(1)), []).append(float(m.group(2)))
    res = {"_note": "tools/ta_roof.sh + tools/ta_roof.py; lines = TCP_TOTAL_CACHE_ACCESSES (cache-line tag accesses, hits and misses); "
                    "peak = csrc/lab/ta_lab map 3 (every CU streams whole lines, 1 KB contiguous per wave-instruction): lines of the PMC "
                    "pass / wall time of the un-profiled run", "lab": {}}
    peak = None
    for k, cs in sorted(lab.items()):
        mp = int(re.search(r"k_ta<(\d)", k).group(1)) if re.search(r"k_ta<(\d)", k) else None
        if mp is None or "TCP_TOTAL_CACHE_ACCESSES_sum" not in cs:
            continue
        long_ids = sorted(lab_dur[k], key=lambda d: -lab_dur[k][d])[:3]
        def mean(c):
            v = [cs[c][d] for d in long_ids if d in cs.get(c, {})]
            return sum(v) / len(v) if v else None
        lines = mean("TCP_TOTAL_CACHE_ACCESSES_sum")
        ninstr = 8.0 * 400 * 8 * 256                      # wave-instructions per timed launch (ta_lab.hip)
        us = sorted(wall.get(mp, [0]))[len(wall.get(mp, [0])) // 2]
        us_prof = sum(lab_dur[k][d] for d in long_ids) / len(long_ids)
        gui = mean("GRBM_GUI_ACTIVE")
        e = dict(lines_per_launch=lines, lines_per_wave_instruction=lines / ninstr, us_wall=us, us_profiled=us_prof,
                 lines_per_s=lines / (us * 1e-6) if us else None, ta_busy_cycles_per_cu=(mean("TA_TA_BUSY_sum") or 0) / 256,
                 ta_buffer_cycles_per_cu=(mean("TA_BUFFER_TOTAL_CYCLES_sum") or 0) / 256,
                 clock_ghz_profiled=gui / 8 / (us_prof * 1e3) if gui else None,
                 tcp_tcc_read_req=mean("TCP_TCC_READ_REQ_sum"), tcp_total_accesses=mean("TCP_TOTAL_ACCESSES_sum"))
        res["lab"][f"map{mp}"] = e
        if mp == 3:
            peak = e
    if peak:
        res["peak_lines_per_s"] = peak["lines_per_s"]
        res["line_bytes"] = round(1024.0 / peak["lines_per_wave_instruction"])
        res["clock_ghz"] = peak["clock_ghz_profiled"]
    b1, d1 = pmc(out / "bench_pmc1")
    b2, _ = pmc(out / "bench_pmc2")
    wl = {}
    for name, pat in (("corr_fwd", "k_corr_fwd_seg"), ("corr_bwd", "k_corr_bwd_strip8")):
        for k, cs in b1.items():
            if k.startswith(pat) and "TCP_TOTAL_CACHE_ACCESSES_sum" in cs:
                v = cs["TCP_TOTAL_CACHE_ACCESSES_sum"]
                wl[name] = sum(v.values()) / len(v)
                extra = {c: sum(x.values()) / len(x) for c, x in cs.items()}
                extra.update({c: sum(x.values()) / len(x) for c, x in b2.get(k, {}).items()})
                extra["us_profiled"] = sum(d1[k].values()) / len(d1[k])
                extra["launches"] = len(v)
                res.setdefault("kernels_detail", {})[k] = extra
    res["corr_B8_C256_38x63_d8"] = wl
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
