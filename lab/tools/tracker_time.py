#!/usr/bin/env python3
"""Tracker glue alone (three correlations + cat + ROIPool: unfused, fused exact, fused FAST), 40 iterations each -- the A/B partner of
D2T_BAND_LEVEL_LAUNCHES=1 in a scan build (one launch per level against one launch for all levels)."""
import sys, json
sys.path.insert(0, "."); sys.path.insert(0, "detect-to-track_amd")
import bench_ops
for e in bench_ops.measure_tracker("cuda:0", 0, 40):
    print(e["op"], e.get("direction") or e.get("variant") or "", e.get("us"), flush=True)
