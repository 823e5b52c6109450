#!/usr/bin/env python3
"""ROIPool backward alone at config 3 (R = 300, C = 1024, 38 x 63, k = 7) -- the program lab/tools/pmc_roi.sh profiles."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT))
from detect_to_track.models import _ext  # noqa: E402
from bench_ops import random_rois, timed  # noqa: E402
dev = "cuda:0"
R, C, H, W = 300, 1024, 38, 63
rois = torch.from_numpy(random_rois(R, 0)).to(dev)
gs = [torch.rand(R, C, 7, 7, device=dev) for _ in range(4)]
fm = torch.rand(C, H, W, device=dev)
print("bwd us", round(timed(lambda i: _ext.roipool_backward(gs[i % 4], rois, H, W), 30, 4), 1))
print("fwd us", round(timed(lambda i: _ext.roipool_forward(fm, rois, 7), 30, 1), 1))
