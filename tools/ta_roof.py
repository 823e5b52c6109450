#!/usr/bin/env python3
"""Reduce the passes of tools/ta_roof.sh to profiles/ta_roof.json (what bench.py's roofline_ta reads).

A "line" here is one request between a CU's vector L1 and the XCD's L2: TCP_TCC_READ_REQ (one per 128-byte line an instruction
misses) + TCP_TCC_WRITE_REQ (one per contiguous run inside a 64-byte sector).  lab/csrc/ta_lab calibrates both the unit and the
peak: with every CU streaming, the request rate is the SAME for 1 KB contiguous per wave-instruction (8 requests) and for 16
channels x 64 bytes at 4-byte alignment (23.5 requests) -- about 2.8 cycles per request and CU -- while the tag-access counter
(TCP_TOTAL_CACHE_ACCESSES) follows the lane order, not the time.  peak_lines_per_s = that rate (requests of the PMC pass / wall time
of the un-profiled run of the same binary); per kernel of the metric workload: requests per launch from the PMC pass of bench.py."""
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
    res = {"_note": "tools/ta_roof.sh + tools/ta_roof.py; lines = L1<->L2 requests per launch = TCP_TCC_READ_REQ (128-byte line fills) + "
                    "TCP_TCC_WRITE_REQ; peak = the request rate lab/csrc/ta_lab sustains with every CU streaming (map 3: 1 KB contiguous "
                    "per wave-instruction; maps 0-2 reach the same rate): requests of the PMC pass / wall time of the un-profiled run",
           "lab": {}}
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
        req = mean("TCP_TCC_READ_REQ_sum")
        e = dict(tag_accesses_per_wave_instruction=lines / ninstr, requests_per_launch=req,
                 requests_per_wave_instruction=req / ninstr if req else None, us_wall=us, us_profiled=us_prof,
                 lines_per_s=req / (us * 1e-6) if us and req else None, ta_busy_cycles_per_cu=(mean("TA_TA_BUSY_sum") or 0) / 256,
                 ta_buffer_cycles_per_cu=(mean("TA_BUFFER_TOTAL_CYCLES_sum") or 0) / 256,
                 clock_ghz_profiled=gui / 8 / (us_prof * 1e3) if gui else None,
                 tcp_tcc_read_req=mean("TCP_TCC_READ_REQ_sum"), tcp_total_accesses=mean("TCP_TOTAL_ACCESSES_sum"))
        res["lab"][f"map{mp}"] = e
        if e["lines_per_s"] and (peak is None or e["lines_per_s"] > peak["lines_per_s"]):
            peak, res["peak_from"] = e, f"map{mp}"
    if peak:
        res["peak_lines_per_s"] = peak["lines_per_s"]
        res["line_bytes"] = round(1024.0 / res["lab"]["map3"]["requests_per_wave_instruction"]) if "map3" in res["lab"] else None
        res["cycles_per_request_per_cu"] = peak["clock_ghz_profiled"] * 1e9 / (peak["lines_per_s"] / 256)
        res["clock_ghz"] = peak["clock_ghz_profiled"]
    b1, d1 = pmc(out / "bench_pmc1")
    b2, _ = pmc(out / "bench_pmc2")
    wl = {}
    for name, pat in (("corr_fwd", "k_corr_fwd_seg"), ("corr_bwd", "k_corr_bwd_strip8")):
        for k, cs in b1.items():
            if k.startswith(pat) and "TCP_TCC_READ_REQ_sum" in b2.get(k, {}):
                v = cs["TCP_TOTAL_CACHE_ACCESSES_sum"]
                extra = {c: sum(x.values()) / len(x) for c, x in cs.items()}
                extra.update({c: sum(x.values()) / len(x) for c, x in b2.get(k, {}).items()})
                wl[name] = extra["TCP_TCC_READ_REQ_sum"] + extra.get("TCP_TCC_WRITE_REQ_sum", 0.0)
                extra["us_profiled"] = sum(d1[k].values()) / len(d1[k])
                extra["launches"] = len(v)
                res.setdefault("kernels_detail", {})[k] = extra
    res["corr_B8_C256_38x63_d8"] = wl
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
