"""pytest configuration: the `gpu` marker, import paths, and the shared helpers.

  -m "not gpu"  runs here (no GPU): oracle vs golden fixtures, host logic, C-ABI symbol export,
                gloo world_size-2 sharding.
  -m gpu        runs on the MI355X box: the parity tests proper, all through the C ABI.

Only tests (and smoke()/bench's cpu_baseline) may touch oracle/.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "detect-to-track_amd"
for p in (str(PKG), str(ROOT / "oracle"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix):
    """Full-tensor fixtures of one op (the *_subsample fixtures have their own schema and test)."""
    return sorted(f for f in GOLDEN.glob(f"{prefix}_*.npz") if not f.stem.endswith("_subsample"))


def golden_ids(prefix):
    return [f.stem for f in golden_files(prefix)]


def load_golden(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def ref_modules():
    """The reference's own kernels (oracle/_ref/, built by oracle/ref_build/Makefile).  GPU-only;
    tests that use them skip when the prebuilt modules are absent."""
    import torch  # noqa: F401  (the extension modules link libtorch)
    ref_dir = ROOT / "oracle" / "_ref"
    if str(ref_dir) not in sys.path:
        sys.path.insert(0, str(ref_dir))
    try:
        import d2t_ref_corr, d2t_ref_roipool, d2t_ref_psroipool
    except ImportError as e:  # pragma: no cover
        pytest.skip(f"oracle/_ref not built: {e}")
    return d2t_ref_corr, d2t_ref_roipool, d2t_ref_psroipool


def random_rois(R, seed, dtype=np.float32):
    """centre U(0.15,0.85)^2, size U(0.05,0.6)^2 (ijhw, fractional)."""
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(0.15, 0.85, (R, 2)), rng.uniform(0.05, 0.6, (R, 2))], 1).astype(dtype)


ADVERSARIAL_ROIS = [
    [0.5, 0.5, 0.5, 0.5], [0.1, 0.1, 0.2, 0.3], [0.95, 0.9, 0.3, 0.4], [0.5, 0.5, 1.0, 1.0],
    [0.5, 0.5, 2.0, 2.0], [0.3, 0.7, 0.0, 0.0], [1.5, 1.5, 0.2, 0.2], [-0.5, -0.5, 0.2, 0.2],
    [0.25, 0.75, 0.01, 0.9], [0.5, 0.5, 0.123, 0.987], [3.0, 3.0, 0.5, 0.5],
]

# RoIs of negative height / width: their bins run in reverse order and one-pixel
# bins stay non-empty in the reference (roipool_cuda.cu:41-50), so forward values are finite and
# the backward deposits gradient.
NEGATIVE_ROIS = [
    [0.5, 0.5, -0.1, 0.4], [0.5, 0.5, 0.3, -0.2], [0.4, 0.6, -0.3, -0.3], [0.2, 0.8, -0.05, 0.5],
    [0.7, 0.3, 0.6, -0.01], [0.5, 0.5, -1.5, 0.5],
]
