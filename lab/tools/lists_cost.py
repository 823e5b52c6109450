#!/usr/bin/env python3
"""Pooling backward just outside the tuned envelope (lab helper for d2t_pool_lists.hip): time per shape."""
import sys
import warnings
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
sys.path.insert(0, str(ROOT))
from detect_to_track.models import _ext  # noqa: E402
from bench_ops import random_rois, timed  # noqa: E402
warnings.simplefilter("ignore")
dev = "cuda:0"
for (R, C, H, W, k) in [(300, 1024, 38, 63, 6), (300, 1024, 38, 63, 14), (3000, 256, 38, 63, 3)]:
    rois = torch.from_numpy(random_rois(R, 0)).to(dev)
    g = torch.rand(R, C, k, k, device=dev)
    print("roipool", (R, C, H, W, k), "bwd", round(timed(lambda i: _ext.roipool_backward(g, rois, H, W), 10, 1), 1), flush=True)
for (R, nT, H, W, k) in [(300, 21, 38, 63, 6), (3000, 31, 38, 63, 6)]:
    rois = torch.from_numpy(random_rois(R, 0)).to(dev)
    g = torch.rand(R, nT, k, k, device=dev)
    print("psroipool", (R, nT, H, W, k), "bwd", round(timed(lambda i: _ext.ps_roipool_backward(g, rois, H, W), 5, 1), 1), flush=True)
