#!/usr/bin/env python3
"""Fold measured HBM bytes into profiles/traffic.json (what bench.py's roofline.traffic and bench_ops.py's ops[].hbm read):
    python3 tools/merge_traffic.py <ops_pmc dir> [<pmc_summary.json of tools/pmc_bwd.sh>] [source tag]
"ops": per op / shape / direction from tools/ops_pmc.sh; the metric workload's two kernels from the pmc summary, if given."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tfile = ROOT / "profiles" / "traffic.json"
t = json.loads(tfile.read_text())
ops = json.loads((Path(sys.argv[1]) / "traffic_ops.json").read_text())
t["ops"] = ops["ops"]
t["_note_ops"] = ops["_note_ops"]
tag = sys.argv[3] if len(sys.argv) > 3 else ""
if len(sys.argv) > 2 and Path(sys.argv[2]).exists():
    s = json.loads(Path(sys.argv[2]).read_text())
    w = t.setdefault("corr_B8_C256_38x63_d8", {})
    for key, pat in (("corr_fwd", "k_corr_fwd_seg"), ("corr_bwd", "k_corr_bwd_strip8")):
        for name, v in s.items():
            if pat in name and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                f, wr = v["FETCH_SIZE"], v["WRITE_SIZE"]
                f = f["mean"] if isinstance(f, dict) else f
                wr = wr["mean"] if isinstance(wr, dict) else wr
                w[key] = round((2.0 * f + wr) * 1024.0)
                break
    if tag:
        t["_note"] = t["_note"].split(" Source:")[0] + f" Source: profiles/{tag}_pmc_summary.json"
tfile.write_text(json.dumps(t, indent=1) + "\n")
print(json.dumps({k: v for k, v in t.items() if k.startswith("corr_")}))
