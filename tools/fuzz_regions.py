"""dev helper: randomised sweep of d2t_region_filter_f32 against the numpy pipeline (the bars of tests/test_region_filter.py)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT / "tests"))
import test_region_filter as T
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for it in range(N):
    h, w = int(rng.integers(1, 60)), int(rng.integers(1, 100))
    k = int(rng.choice([1, 2, 7, 16, 63, 64, 65, 128, 300, 511, 512, 513, 1000, 1024, 1025, 3000, 4096]))
    case = (h, w, float(rng.choice([0.0, 0.05, 0.3, 0.7, 0.95, 0.999])), k, float(rng.choice([0.0, 0.3, 0.5, 0.7, 0.95])), float(rng.choice([0.1, 0.5, 1.0])))
    try:
        T.test_matches_numpy_pipeline(case)
    except Exception as e:
        bad += 1; print("REGION FAIL", case, str(e)[:300], flush=True)
    if it % 25 == 24: print(f"{it + 1} cases, {bad} failures", flush=True)
print("done", N, "cases,", bad, "failures")
