"""dev helper: long randomised sweep of the second kernel tier (what the default dispatch runs OUTSIDE the tuned envelope) against
the thread-per-element anchor kernels (D2T_IMPL_GENERIC).  Bars: every backward bit for bit (NaN pattern included); ROIPool forward
(summed-area tables, k <= 16) NaN pattern exact and values within 1e-5; correlation forward bit for bit.
usage: python tools/fuzz_tier2.py [seed] [iterations]"""
import sys, warnings, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT / "tests"))
from detect_to_track.models import _ext
from test_tuned_vs_generic_fuzz import _rois, TOL
warnings.simplefilter("ignore")
DEV = "cuda:0"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0


def same(a, b):
    return torch.equal(a.isnan(), b.isnan()) and torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))


for it in range(N):
    kind = it % 3
    try:
        if kind == 0:                                                 # ROIPool, any k
            R, C, H, W = int(rng.integers(1, 700)), int(rng.integers(1, 200)), int(rng.integers(1, 60)), int(rng.integers(1, 129))
            k = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 12, 14, 16, 17, 20]))
            shape = ("roipool", R, C, H, W, k)
            dt = torch.float64 if it % 9 == 0 else torch.float32
            fm = torch.from_numpy(rng.standard_normal((C, H, W))).to(dt).to(DEV)
            gout = torch.from_numpy(rng.standard_normal((R, C, k, k))).to(dt).to(DEV)
            rois = _rois(rng, R).to(dt)
            a, b = _ext.roipool_forward(fm, rois, k, 0), _ext.roipool_forward(fm, rois, k, 1)
            assert torch.equal(a.isnan(), b.isnan()), "forward NaN pattern"
            if dt == torch.float32:
                torch.testing.assert_close(torch.nan_to_num(a), torch.nan_to_num(b), **TOL)
            else:
                assert same(a, b), "f64 forward"
            assert same(_ext.roipool_backward(gout, rois, H, W, 0), _ext.roipool_backward(gout, rois, H, W, 1)), "backward bits"
        elif kind == 1:                                               # PSROIPool, k != 7
            R, nT, H, W = int(rng.integers(1, 700)), int(rng.integers(1, 33)), int(rng.integers(1, 60)), int(rng.integers(1, 129))
            k = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 12]))
            shape = ("psroipool", R, nT, H, W, k)
            dt = torch.float64 if it % 10 == 1 else torch.float32
            fm = torch.from_numpy(rng.standard_normal((nT * k * k, H, W))).to(dt).to(DEV)
            gout = torch.from_numpy(rng.standard_normal((R, nT, k, k))).to(dt).to(DEV)
            rois = _rois(rng, R).to(dt)
            assert same(_ext.ps_roipool_forward(fm, rois, nT, k, 0), _ext.ps_roipool_forward(fm, rois, nT, k, 1)), "forward bits"
            assert same(_ext.ps_roipool_backward(gout, rois, H, W, 0), _ext.ps_roipool_backward(gout, rois, H, W, 1)), "backward bits"
        else:                                                         # correlation outside d_max = 8 / stride 1
            d, s = int(rng.integers(0, 17)), int(rng.choice([1, 1, 2, 3]))
            if d == 8 and s == 1:
                d = 7
            B, C, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 150)), int(rng.integers(1, 40)), int(rng.integers(4, 70))
            if (2 * d + 1) ** 2 * B * H * W > 4_000_000:
                H = max(1, 4_000_000 // ((2 * d + 1) ** 2 * B * W))
            shape = ("corr", B, C, H, W, d, s)
            f0 = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).to(DEV)
            f1 = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).to(DEV)
            g = torch.from_numpy(rng.standard_normal((B, H, W, 2 * d + 1, 2 * d + 1)).astype(np.float32)).to(DEV)
            assert torch.equal(_ext.pointwise_correlation_forward(f0, f1, d, s, 0), _ext.pointwise_correlation_forward(f0, f1, d, s, 1)), "forward bits"
            a0, a1 = _ext.pointwise_correlation_backward(g, f0, f1, d, s, 0)
            b0, b1 = _ext.pointwise_correlation_backward(g, f0, f1, d, s, 1)
            assert torch.equal(a0, b0) and torch.equal(a1, b1), "backward bits"
    except Exception as e:
        bad += 1
        print("FAIL", shape, str(e)[:300], flush=True)
    if it % 25 == 24:
        print(f"{it + 1} / {N}, failures so far: {bad}", flush=True)
print("second-tier fuzz:", N, "cases,", bad, "failures")
sys.exit(1 if bad else 0)
