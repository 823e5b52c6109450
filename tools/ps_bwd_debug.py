#!/usr/bin/env python3
"""dev helper: where does the tuned PSROIPool backward differ from the generic kernel on a golden fixture?"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _ext
dev = "cuda:0"
g = np.load(ROOT / "tests" / "golden" / (sys.argv[1] if len(sys.argv) > 1 else "psroipool_adv_n1_k7_38x63_f32.npz"))
nT, k = int(g["nT"]), int(g["k"])
_, H, W = g["fm"].shape
gout, rois = torch.from_numpy(g["gout"]).to(dev), torch.from_numpy(g["rois"]).to(dev)
print("R", rois.shape[0], "nT", nT, "gout finite", bool(torch.isfinite(gout).all()))
print(g["rois"])
a = _ext.ps_roipool_backward(gout, rois, H, W, 0).cpu().numpy()
b = _ext.ps_roipool_backward(gout, rois, H, W, 1).cpu().numpy()
bad = ~np.isclose(a, b, rtol=1e-5, atol=1e-5, equal_nan=True)
print("mismatches", bad.sum(), "of", bad.size)
ch, ys, xs = np.nonzero(bad)
for c in np.unique(ch)[:12]:
    m = ch == c
    print(f"channel {c}: rows {sorted(set(ys[m]))[:20]} cols {min(xs[m])}..{max(xs[m])}  e.g. tuned {a[c, ys[m][0], xs[m][0]]:.5f} generic {b[c, ys[m][0], xs[m][0]]:.5f}")
