#!/usr/bin/env python3
"""Correlation just outside the tuned envelope: forward / backward time per shape (lab helper for d2t_corr_blocked.hip)."""
import sys
import warnings
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
sys.path.insert(0, str(ROOT))
from detect_to_track.models import _ext  # noqa: E402
from bench_ops import timed  # noqa: E402
warnings.simplefilter("ignore")
dev = "cuda:0"
shapes = [(8, 256, 38, 63, 7, 1), (1, 256, 38, 63, 7, 1), (8, 256, 38, 63, 4, 1), (8, 256, 38, 63, 12, 1), (8, 256, 38, 63, 8, 2),
          (8, 256, 38, 63, 2, 1), (8, 256, 38, 63, 5, 1), (8, 256, 38, 63, 6, 1), (2, 1024, 38, 63, 6, 1)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (B, C, H, W, d, s) in shapes:
    f0, f1 = torch.rand(B, C, H, W, device=dev), torch.rand(B, C, H, W, device=dev)
    g = torch.rand(B, H, W, 2 * d + 1, 2 * d + 1, device=dev)
    tf = timed(lambda i: _ext.pointwise_correlation_forward(f0, f1, d, s), 10, 1)
    tb = timed(lambda i: _ext.pointwise_correlation_backward(g, f0, f1, d, s), 10, 1)
    print((B, C, H, W, d, s), "fwd", round(tf, 1), "bwd", round(tb, 1), flush=True)
