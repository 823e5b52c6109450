#!/usr/bin/env python3
"""ROIPool backward at config 3 under the scan build's D2T_ROI_CFG (c-tiles per task, waves per task): 1 (1,4) | 2 (2,4) default | 3 (4,4) | 4 (2,2).
The knob is read once per process: run once per value."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
import bench_ops  # noqa: E402
st = torch.cuda.current_stream().cuda_stream
for shape in ((300, 1024, 38, 63), (300, 1891, 38, 75)):
    e = bench_ops.measure_roipool("cuda:0", *shape, 0, 40, st)
    print(shape, "bwd", round(e[1]["us"], 1), flush=True)
