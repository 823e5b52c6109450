#!/usr/bin/env python3
"""Tracker's three correlations (B = 1, C = 512 / 1024 / 2048, 38 x 75): separate calls against the fused levels call, exact (AUTO)
and FAST, forward and backward."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT))
from detect_to_track.models import _ext, _native  # noqa: E402
from bench_ops import timed  # noqa: E402
dev = "cuda:0"
H, W, Cs = 38, 75, (512, 1024, 2048)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = int(sys.argv[2]) if len(sys.argv) > 2 else W
print("B =", B, "W =", W, flush=True)
f0 = [torch.rand(B, C, H, W, device=dev) for C in Cs]
f1 = [torch.rand(B, C, H, W, device=dev) for C in Cs]
g = torch.rand(B, len(Cs) * 289, H, W, device=dev)
gs = [torch.rand(B, H, W, 17, 17, device=dev) for _ in Cs]
for C, a, b in zip(Cs, f0, f1):
    print("separate fwd C", C, round(timed(lambda i: _ext.pointwise_correlation_forward(a, b, 8, 1), 20, 1), 1), flush=True)
print("separate fwd all", round(timed(lambda i: [_ext.pointwise_correlation_forward(a, b, 8, 1) for a, b in zip(f0, f1)], 20, 1), 1))
for name, impl in (("AUTO", 0), ("FAST", _native.IMPL_FAST)):
    print("levels fwd", name, round(timed(lambda i: _ext.pointwise_correlation_levels_forward(f0, f1, 8, 1, impl=impl), 20, 1), 1), flush=True)
print("separate bwd all", round(timed(lambda i: [_ext.pointwise_correlation_backward(gg, a, b, 8, 1) for gg, a, b in zip(gs, f0, f1)], 20, 1), 1))
print("levels bwd", round(timed(lambda i: _ext.pointwise_correlation_levels_backward(g, 0, f0, f1, 8, 1), 20, 1), 1), flush=True)
