#!/usr/bin/env python3
"""LOST experiment of round 5 (kept for the record; the kernels and the knobs D2T_ROI_FEW_MAXR / D2T_ROI_FEW_OFF it drove are no longer in the library:
lab/csrc/roipool_fwd_few.inc, profiles/r05_e_roipool_few_rois_lost.txt).  ROIPool forward at few RoIs (the tracker's eval path): the boxes-through-LDS kernel against the reference-order kernel (bit for bit)
and against the summed-area kernel (time).  Scan build (D2T_ROI_FEW_MAXR / D2T_ROI_FEW_OFF are read per call):
    D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/lib_knobs/libd2t_ops.so python3 tools/roipool_small_r.py"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
import bench_ops  # noqa: E402
from detect_to_track.models import _ext  # noqa: E402

dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
os.environ["D2T_ROI_FEW_MAXR"] = "100000"
bad = 0
rng = np.random.default_rng(7)
for (R, C, H, W, k) in [(8, 1891, 38, 75, 7), (5, 33, 21, 40, 7), (3, 17, 38, 75, 5), (9, 40, 16, 16, 6), (4, 20, 100, 120, 7), (2, 7, 9, 11, 16), (31, 64, 38, 63, 7), (1, 1, 1, 1, 1)]:
    fm = torch.randn(C, H, W, device=dev)
    rois = np.concatenate([rng.uniform(-0.1, 1.1, (R, 2)), rng.uniform(-0.05, 1.3, (R, 2))], 1).astype(np.float32)
    if R > 2:
        rois[1] = [0.5, 0.5, 0.0, 0.0]; rois[2] = [0.3, float("nan"), 0.2, 0.2]
    rois_t = torch.from_numpy(rois).to(dev)
    if C > 3:
        fm[1, H // 2, W // 2] = float("inf"); fm[2, 0, 0] = float("nan")
    got = _ext.roipool_forward(fm, rois_t, k, 0)
    want = _ext.roipool_forward(fm, rois_t, k, 1)
    same = torch.equal(torch.nan_to_num(got, nan=12345.0), torch.nan_to_num(want, nan=12345.0)) and torch.equal(torch.isnan(got), torch.isnan(want))
    bad += 0 if same else 1
    print(f"check R={R} C={C} {H}x{W} k={k}: {'bit-identical' if same else 'MISMATCH'}", flush=True)
for R in (8, 16, 32, 64, 128):
    row = []
    for name, env in (("row-batched loads", {"D2T_ROI_FEW_MAXR": "100000", "D2T_ROI_FEW_OFF": "0"}), ("thread per output", {"D2T_ROI_FEW_MAXR": "100000", "D2T_ROI_FEW_OFF": "1"}),
                      ("summed-area", {"D2T_ROI_FEW_MAXR": "0", "D2T_ROI_FEW_OFF": "0"})):
        os.environ.update(env)
        e = bench_ops.measure_roipool(dev, R, 1891, 38, 75, 0, 30, st)
        row.append(f"{name} {e[0]['us']:.1f}")
    print(f"R={R} C=1891 38x75 fwd [us]: " + " | ".join(row), flush=True)
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
