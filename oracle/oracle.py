"""numpy front-end of the CPU oracle (oracle/libd2t_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg -- never by the product package (detect-to-track_amd/), which has no CPU path.

Every function takes / returns C-contiguous numpy arrays of dtype float32 or float64 and
follows the reference kernel cited in oracle/d2t_oracle_impl.h.
"""
import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libd2t_oracle.so"


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (a few seconds)."""
    src_newer = _LIB_PATH.exists() and any(
        (_HERE / f).stat().st_mtime > _LIB_PATH.stat().st_mtime for f in ("d2t_oracle.c", "d2t_oracle_impl.h"))
    if force or src_newer or not _LIB_PATH.exists():
        subprocess.run(["make", "-C", str(_HERE), "-B", "libd2t_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
    return _lib


def _sfx(a: np.ndarray) -> str:
    if a.dtype == np.float32:
        return "f32"
    if a.dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {a.dtype}")


def _c(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a)


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def set_threads(n: int) -> None:
    """Number of OpenMP threads of the following calls.  The environment variable only counts before
    libgomp's first parallel region, so the runtime of the already-loaded library is told as well."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    lib()                                              # maps libgomp (DT_NEEDED of the oracle)
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:                                    # pragma: no cover
        pass


def corr_fwd(fm0, fm1, d, s):
    fm0, fm1 = _c(fm0), _c(fm1)
    B, C, H, W = fm0.shape
    cw = 2 * d + 1
    out = np.empty((B, H, W, cw, cw), dtype=fm0.dtype)
    getattr(lib(), f"d2t_oracle_corr_fwd_{_sfx(fm0)}")(_p(fm0), _p(fm1), _p(out), B, C, H, W, d, s)
    return out


def corr_bwd(gout, fm0, fm1, d, s):
    gout, fm0, fm1 = _c(gout), _c(fm0), _c(fm1)
    B, C, H, W = fm0.shape
    g0, g1 = np.empty_like(fm0), np.empty_like(fm1)
    getattr(lib(), f"d2t_oracle_corr_bwd_{_sfx(fm0)}")(_p(gout), _p(fm0), _p(fm1), _p(g0), _p(g1), B, C, H, W, d, s)
    return g0, g1


def corr_mask(H, W, d, s):
    cw = 2 * d + 1
    m = np.empty((H, W, cw, cw), dtype=np.uint8)
    lib().d2t_oracle_corr_mask(_p(m), H, W, d, s)
    return m


def roipool_fwd(fm, rois, k):
    fm, rois = _c(fm), _c(rois)
    C, H, W = fm.shape
    R = rois.shape[0]
    out = np.empty((R, C, k, k), dtype=fm.dtype)
    getattr(lib(), f"d2t_oracle_roipool_fwd_{_sfx(fm)}")(_p(fm), _p(rois), _p(out), R, C, H, W, k)
    return out


def roipool_bwd(gout, rois, H, W):
    gout, rois = _c(gout), _c(rois)
    R, C, k, _ = gout.shape
    gin = np.empty((C, H, W), dtype=gout.dtype)
    getattr(lib(), f"d2t_oracle_roipool_bwd_{_sfx(gout)}")(_p(gout), _p(rois), _p(gin), R, C, H, W, k)
    return gin


def roipool_bins(rois, H, W, k, position_sensitive=False):
    rois = _c(rois)
    R = rois.shape[0]
    out = np.empty((R, k, k, 4), dtype=np.int32)
    name = "psroipool_bins" if position_sensitive else "roipool_bins"
    getattr(lib(), f"d2t_oracle_{name}_{_sfx(rois)}")(_p(rois), _p(out), R, H, W, k)
    return out


def psroipool_fwd(fm, rois, nT, k):
    fm, rois = _c(fm), _c(rois)
    _, H, W = fm.shape
    R = rois.shape[0]
    out = np.empty((R, nT, k, k), dtype=fm.dtype)
    getattr(lib(), f"d2t_oracle_psroipool_fwd_{_sfx(fm)}")(_p(fm), _p(rois), _p(out), R, nT, H, W, k)
    return out


def psroipool_bwd(gout, rois, H, W):
    gout, rois = _c(gout), _c(rois)
    R, nT, k, _ = gout.shape
    gin = np.empty((nT * k * k, H, W), dtype=gout.dtype)
    getattr(lib(), f"d2t_oracle_psroipool_bwd_{_sfx(gout)}")(_p(gout), _p(rois), _p(gin), R, nT, H, W, k)
    return gin


def psroipool_channels(nT, k):
    ch = np.empty((nT, k, k), dtype=np.int32)
    lib().d2t_oracle_psroipool_channels(_p(ch), nT, k)
    return ch


# ---- wide-accumulator yardsticks (terms as the reference forms them, summed in double, rounded once):
# each returns (gradient(s), magnitude(s)) with magnitude = sum |term| per element in float64.  A kernel's
# f32 gradient is held to  |got - ref| <= 1e-5 * magnitude  (plus one ulp of the result).
def roipool_bwd_acc64(gout, rois, H, W):
    gout, rois = _c(gout), _c(rois)
    R, C, k, _ = gout.shape
    gin = np.empty((C, H, W), dtype=gout.dtype)
    mag = np.empty((C, H, W), dtype=np.float64)
    getattr(lib(), f"d2t_oracle_roipool_bwd_acc64_{_sfx(gout)}")(_p(gout), _p(rois), _p(gin), _p(mag), R, C, H, W, k)
    return gin, mag


def psroipool_bwd_acc64(gout, rois, H, W):
    gout, rois = _c(gout), _c(rois)
    R, nT, k, _ = gout.shape
    gin = np.empty((nT * k * k, H, W), dtype=gout.dtype)
    mag = np.empty((nT * k * k, H, W), dtype=np.float64)
    getattr(lib(), f"d2t_oracle_psroipool_bwd_acc64_{_sfx(gout)}")(_p(gout), _p(rois), _p(gin), _p(mag), R, nT, H, W, k)
    return gin, mag


def corr_bwd_acc64(gout, fm0, fm1, d, s):
    gout, fm0, fm1 = _c(gout), _c(fm0), _c(fm1)
    B, C, H, W = fm0.shape
    g0, g1 = np.empty_like(fm0), np.empty_like(fm1)
    m0, m1 = np.empty(fm0.shape, dtype=np.float64), np.empty(fm0.shape, dtype=np.float64)
    getattr(lib(), f"d2t_oracle_corr_bwd_acc64_{_sfx(fm0)}")(_p(gout), _p(fm0), _p(fm1), _p(g0), _p(g1), _p(m0), _p(m1),
                                                              B, C, H, W, d, s)
    return (g0, g1), (m0, m1)


def assert_within_contract(got, ref, mag, rel=1e-5, what="gradient"):
    """|got - ref| <= rel * sum|terms| + one float32 ulp of the reference, element-wise (NaN patterns must match)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN pattern differs"
    fin = np.isfinite(ref)
    tol = rel * np.asarray(mag)[fin] + np.spacing(np.abs(ref[fin]).astype(np.float32)).astype(np.float64)
    err = np.abs(got[fin] - ref[fin])
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())} elements beyond {rel:g} * sum|terms|; worst err/tol = "
                           f"{float((err / np.maximum(tol, 1e-300)).max()):.3g}")
