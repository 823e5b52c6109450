"""dev helper: long randomised sweep of the pooling kernels, tuned vs generic (same bars as tests/test_tuned_vs_generic_fuzz.py)."""
import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT / "tests"))
from detect_to_track.models import _ext
from test_tuned_vs_generic_fuzz import _rois, TOL, GENERIC, TUNED
DEV = "cuda:0"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
for it in range(N):
    R = int(rng.integers(1, 700)) if it % 5 else int(rng.integers(1400, 5000))     # every fifth case: thousands of RoIs (PSROIPool row form with RoI ranges)
    H = int(rng.integers(1, 60)); W = int(rng.integers(1, 129))
    if it % 2 == 0:
        C = int(rng.integers(1, 200))
        fm = torch.from_numpy(rng.random((C, H, W), dtype=np.float32)).to(DEV)
        gout = torch.from_numpy(rng.standard_normal((R, C, 7, 7)).astype(np.float32)).to(DEV)
        rois = _rois(rng, R)
        try:
            a, b = _ext.roipool_forward(fm, rois, 7, TUNED), _ext.roipool_forward(fm, rois, 7, GENERIC)
            assert torch.equal(a.isnan(), b.isnan())
            torch.testing.assert_close(torch.nan_to_num(a), torch.nan_to_num(b), **TOL)
            if H * W * 32 + 160 * 36 <= 160 * 1024:             # round 6: the direct kernel (8 interleaved planes in LDS) -- the reference's bits
                assert torch.equal(torch.nan_to_num(a).view(torch.int32), torch.nan_to_num(b).view(torch.int32)), "ROIPool forward k = 7 not bit-identical"
            # gradients are sums of thousands of signed f32 terms in kernel-specific orders: the yardstick is
            # f32 rounding of the sum of their MAGNITUDES (the backward of |gradOut|), not of the result
            t, g = _ext.roipool_backward(gout, rois, H, W, TUNED), _ext.roipool_backward(gout, rois, H, W, GENERIC)
            mag = _ext.roipool_backward(gout.abs(), rois, H, W, GENERIC)
            assert bool(((t - g).abs() <= 4e-6 * mag + 1e-6).all()), f"max excess {float(((t - g).abs() - 4e-6 * mag).max())}"
        except Exception as e:
            bad += 1; print("ROIPOOL FAIL", (R, C, H, W), str(e)[:300], flush=True)
    else:
        nT = int(rng.integers(1, 40))
        fm = torch.from_numpy(rng.random((nT * 49, H, W), dtype=np.float32)).to(DEV)
        gout = torch.from_numpy(rng.standard_normal((R, nT, 7, 7)).astype(np.float32)).to(DEV)
        rois = _rois(rng, R)
        try:
            assert torch.equal(_ext.ps_roipool_forward(fm, rois, nT, 7, TUNED), _ext.ps_roipool_forward(fm, rois, nT, 7, GENERIC))
            t, g = _ext.ps_roipool_backward(gout, rois, H, W, TUNED), _ext.ps_roipool_backward(gout, rois, H, W, GENERIC)
            mag = _ext.ps_roipool_backward(gout.abs(), rois, H, W, GENERIC)
            assert bool(((t - g).abs() <= 4e-6 * mag + 1e-6).all()), f"max excess {float(((t - g).abs() - 4e-6 * mag).max())}"
        except Exception as e:
            bad += 1; print("PSROIPOOL FAIL", (R, nT, H, W), str(e)[:300], flush=True)
    if it % 25 == 24: print(f"{it + 1} cases, {bad} failures", flush=True)
print("done", N, "cases,", bad, "failures")
