#!/usr/bin/env python3
"""Where does the 8-pixel-strip backward differ from the reference-order kernel?  (developer script)"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _ext
for (B, C, H, W) in ((1, 32, 17, 24), (1, 16, 21, 30), (1, 16, 38, 63)):
    rng = np.random.default_rng(1)
    fm0, fm1 = (torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).cuda() for _ in range(2))
    gout = torch.from_numpy(rng.standard_normal((B, H, W, 17, 17)).astype(np.float32)).cuda()
    r0, r1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, 1)
    for impl in (6, 7):
        g0, g1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, impl)
        for name, g, r in (("g0", g0, r0), ("g1", g1, r1)):
            err = (g - r).abs()
            bad = (err > 1e-3).nonzero()
            pix = sorted(set((int(i), int(j)) for _, _, i, j in bad.tolist()))
            print((B, C, H, W), "impl", impl, name, "max err", float(err.max()), "bad elements", len(bad), "pixels", pix[:40])
