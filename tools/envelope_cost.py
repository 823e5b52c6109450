#!/usr/bin/env python3
"""What leaving the tuned envelope costs (d_max = 8, stride 1, W >= 20 / k = 7 / float32): each op at ONE real shape, tuned
dispatch against the shape just outside it, through the nn.Module-level entry points.  One JSON line per case."""
import json
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _ext  # noqa: E402
sys.path.insert(0, str(ROOT))
from bench_ops import random_rois, timed  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)


def corr(B, C, H, W, d, s, dtype, iters):
    f0, f1 = torch.rand(B, C, H, W, device=dev, dtype=dtype), torch.rand(B, C, H, W, device=dev, dtype=dtype)
    g = torch.rand(B, H, W, 2 * d + 1, 2 * d + 1, device=dev, dtype=dtype)
    tf = timed(lambda i: _ext.pointwise_correlation_forward(f0, f1, d, s), iters, 1)
    tb = timed(lambda i: _ext.pointwise_correlation_backward(g, f0, f1, d, s), iters, 1)
    return tf, tb


def pool(op, R, C, H, W, k, dtype, iters):
    nT = C
    fm = torch.rand((nT * k * k if op == "ps" else C), H, W, device=dev, dtype=dtype)
    rois = torch.from_numpy(random_rois(R, 0)).to(dev).to(dtype)
    if op == "ps":
        out = _ext.ps_roipool_forward(fm, rois, nT, k)
        g = torch.rand_like(out)
        return (timed(lambda i: _ext.ps_roipool_forward(fm, rois, nT, k), iters, 1),
                timed(lambda i: _ext.ps_roipool_backward(g, rois, H, W), iters, 1))
    out = _ext.roipool_forward(fm, rois, k)
    g = torch.rand_like(out)
    return (timed(lambda i: _ext.roipool_forward(fm, rois, k), iters, 1), timed(lambda i: _ext.roipool_backward(g, rois, H, W), iters, 1))


cases = [("corr B8 C256 38x63 d=8 s=1 f32 (tuned)", lambda: corr(8, 256, 38, 63, 8, 1, torch.float32, 20)),
         ("corr B8 C256 38x63 d=7 s=1 f32 (generic)", lambda: corr(8, 256, 38, 63, 7, 1, torch.float32, 3)),
         ("corr B8 C256 38x63 d=8 s=2 f32 (generic)", lambda: corr(8, 256, 38, 63, 8, 2, torch.float32, 3)),
         ("corr B8 C256 38x63 d=8 s=1 f64 (generic)", lambda: corr(8, 256, 38, 63, 8, 1, torch.float64, 3)),
         ("roipool R300 C1024 38x63 k=7 f32 (tuned)", lambda: pool("roi", 300, 1024, 38, 63, 7, torch.float32, 20)),
         ("roipool R300 C1024 38x63 k=6 f32 (generic)", lambda: pool("roi", 300, 1024, 38, 63, 6, torch.float32, 5)),
         ("roipool R300 C1024 38x63 k=7 f64 (generic)", lambda: pool("roi", 300, 1024, 38, 63, 7, torch.float64, 5)),
         ("psroipool R300 nT21 38x63 k=7 f32 (tuned)", lambda: pool("ps", 300, 21, 38, 63, 7, torch.float32, 20)),
         ("psroipool R300 nT21 38x63 k=6 f32 (generic)", lambda: pool("ps", 300, 21, 38, 63, 6, torch.float32, 5)),
         ("psroipool R300 nT21 38x63 k=7 f64 (generic)", lambda: pool("ps", 300, 21, 38, 63, 7, torch.float64, 5))]
for name, fn in cases:
    tf, tb = fn()
    print(json.dumps(dict(case=name, fwd_us=round(tf, 1), bwd_us=round(tb, 1))), flush=True)
