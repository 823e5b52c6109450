"""dev helper: what does ONE Inf in a feature map cost the tuned backward correlation (repair path)?"""
import sys, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT))
from detect_to_track.models import _ext
from bench_ops import timed
B, C, H, W = 8, 256, 38, 63
fm0 = torch.rand(B, C, H, W, device="cuda"); fm1 = torch.rand(B, C, H, W, device="cuda"); go = torch.rand(B, H, W, 17, 17, device="cuda")
us = timed(lambda i: _ext.pointwise_correlation_backward(go, fm0, fm1, 8, 1), 20, 1)
print(f"finite inputs: {us:.1f} us")
fm1[3, 100, 17, 30] = float("inf")
us = timed(lambda i: _ext.pointwise_correlation_backward(go, fm0, fm1, 8, 1), 20, 1)
print(f"one Inf in FM1: {us:.1f} us")
fm0[:] = float("nan")
us = timed(lambda i: _ext.pointwise_correlation_backward(go, fm0, fm1, 8, 1), 5, 1)
print(f"FM0 all NaN: {us:.1f} us")
