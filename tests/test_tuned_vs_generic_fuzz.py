"""Randomised shape sweep: the gfx950-tuned kernels against the type-generic ones, same inputs.

The generic kernels (impl = D2T_IMPL_GENERIC) are pinned to the CPU oracle and to the reference's
fixtures elsewhere in tests/; here they are the yardstick for shapes no fixture has: odd map
sizes, channel counts that leave partial chunks, maps wider than a wave, RoIs thinner than a bin,
maps above the LDS-plane limit of the PSROIPool backward, batch sizes that pick each of the
correlation's forward kernels.  Seeds are fixed: the sweep is the same on every run.

Bars: correlation and PSROIPool forward bit-exact (they keep the reference's summation order);
ROIPool forward (summed-area tables) and every backward |delta| <= 1e-5 abs/rel.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(rtol=1e-5, atol=1e-5)
GENERIC, TUNED = 1, 2         # D2T_IMPL_MFMA: the tuned kernels, demanded (no silent generic dispatch)


def _rois(rng, R):
    """(R,4) ijhw fractions: mostly ordinary boxes, some thin / tiny / oversize / off-map ones."""
    c = rng.uniform(0.0, 1.0, (R, 2))
    s = rng.uniform(0.02, 0.9, (R, 2))
    kind = rng.integers(0, 10, R)
    s[kind == 0] = rng.uniform(0.0, 0.03, (int((kind == 0).sum()), 2))      # thinner than a bin
    s[kind == 1] = rng.uniform(1.0, 2.5, (int((kind == 1).sum()), 2))       # larger than the map
    c[kind == 2] = rng.uniform(-0.5, 1.5, (int((kind == 2).sum()), 2))      # centre off the map
    s[kind == 3, 0] = 0.01                                                  # a sliver
    s[kind == 4] *= -1.0                                                    # negative height and width
    s[kind == 5, 1] *= -0.3                                                 # negative width only
    return torch.from_numpy(np.concatenate([c, s], 1).astype(np.float32)).to(DEV)


def _corr_cases():
    rng = np.random.default_rng(20261003)
    cases = []
    for _ in range(14):
        B = int(rng.integers(1, 5))
        C = int(rng.integers(1, 90))
        H = int(rng.integers(1, 46))
        W = int(rng.integers(20, 82))                                        # tuned kernels need W >= 20
        cases.append((B, C, H, W))
    cases += [(9, 17, 38, 63), (1, 272, 9, 20), (2, 31, 40, 21), (6, 5, 3, 97)]
    return cases


@pytest.mark.parametrize("case", _corr_cases(), ids=str)
def test_correlation_tuned_equals_generic(case):
    from detect_to_track.models import _ext
    B, C, H, W = case
    g = torch.Generator(device="cpu").manual_seed(B * 1000003 + C * 1009 + H * 31 + W)
    fm0 = torch.rand(B, C, H, W, generator=g).to(DEV)
    fm1 = torch.rand(B, C, H, W, generator=g).to(DEV)
    gout = torch.rand(B, H, W, 17, 17, generator=g).to(DEV)
    out_t = _ext.pointwise_correlation_forward(fm0, fm1, 8, 1, TUNED)
    out_g = _ext.pointwise_correlation_forward(fm0, fm1, 8, 1, GENERIC)
    assert torch.equal(out_t, out_g), f"max |delta| {(out_t - out_g).abs().max().item()}"
    # the tuned backward (the 8-wave strip kernel) takes maps from 17 rows up; lower maps are outside its envelope (ABI 1.06): there the
    # default dispatch (second tier, bit-identical to the generic kernels) is what runs, and demanding the tuned path is D2T_EINVAL
    t0, t1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, TUNED if H >= 17 else 0)
    g0, g1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, GENERIC)
    if H < 17:
        with pytest.raises(RuntimeError, match="invalid argument"):
            _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, TUNED)
    torch.testing.assert_close(t0, g0, **TOL)
    torch.testing.assert_close(t1, g1, **TOL)


def _pool_cases():
    rng = np.random.default_rng(7 * 20261003)
    cases = []
    for _ in range(10):
        cases.append((int(rng.integers(1, 70)), int(rng.integers(1, 150)), int(rng.integers(1, 50)), int(rng.integers(1, 90))))
    cases += [(65, 64, 38, 63), (3, 65, 2, 130), (130, 1, 47, 5), (20, 300, 38, 75), (7, 33, 70, 70)]
    # the backward's GEMM form serves maps up to 128 columns (4, 5 or 8 column tiles), the per-pixel kernel the wider ones
    cases += [(10, 40, 12, 128), (9, 33, 10, 140), (257, 17, 1, 77), (300, 31, 3, 16), (1, 1, 1, 1)]
    return cases


@pytest.mark.parametrize("case", _pool_cases(), ids=str)
def test_roipool_tuned_equals_generic(case):
    from detect_to_track.models import _ext
    R, C, H, W = case
    rng = np.random.default_rng(R * 7919 + C * 31 + H * 7 + W)
    fm = torch.from_numpy(rng.random((C, H, W), dtype=np.float32)).to(DEV)
    gout = torch.from_numpy(rng.random((R, C, 7, 7), dtype=np.float32)).to(DEV)
    rois = _rois(rng, R)
    out_t = _ext.roipool_forward(fm, rois, 7, TUNED)
    out_g = _ext.roipool_forward(fm, rois, 7, GENERIC)
    assert torch.equal(out_t.isnan(), out_g.isnan())                # NaN pattern bit-exact
    torch.testing.assert_close(torch.nan_to_num(out_t), torch.nan_to_num(out_g), **TOL)   # summed-area tables: exact sums
    gin_t = _ext.roipool_backward(gout, rois, H, W, TUNED)
    gin_g = _ext.roipool_backward(gout, rois, H, W, GENERIC)
    torch.testing.assert_close(gin_t, gin_g, **TOL)


def _ps_cases():
    rng = np.random.default_rng(13 * 20261003)
    cases = []
    for _ in range(10):
        cases.append((int(rng.integers(1, 90)), int(rng.integers(1, 8)), int(rng.integers(1, 50)), int(rng.integers(1, 90))))
    cases += [(300, 12, 38, 63), (5, 2, 64, 65), (64, 1, 60, 100), (129, 11, 38, 75)]       # incl. maps > 4096 pixels
    for _ in range(8):                                                       # 12..32 targets: the backward's GEMM form
        cases.append((int(rng.integers(1, 400)), int(rng.integers(12, 33)), int(rng.integers(1, 50)), int(rng.integers(1, 129))))
    cases += [(1200, 9, 20, 30), (70, 33, 11, 40), (33, 20, 10, 140)]        # 8..11 targets at R >= 1000; beyond 32 targets / 128 columns: other designs
    cases += [(450, 33, 30, 41), (720, 20, 16, 140), (1500, 40, 12, 20)]     # ... with R * nT >= 14000: the sorted corner lists (default dispatch)
    return cases


@pytest.mark.parametrize("case", _ps_cases(), ids=str)
def test_ps_roipool_tuned_equals_generic(case):
    from detect_to_track.models import _ext
    R, nT, H, W = case
    rng = np.random.default_rng(R * 104729 + nT * 1013 + H * 13 + W)
    fm = torch.from_numpy(rng.random((nT * 49, H, W), dtype=np.float32)).to(DEV)
    gout = torch.from_numpy(rng.random((R, nT, 7, 7), dtype=np.float32)).to(DEV)
    rois = _rois(rng, R)
    out_t = _ext.ps_roipool_forward(fm, rois, nT, 7, TUNED)
    out_g = _ext.ps_roipool_forward(fm, rois, nT, 7, GENERIC)
    assert torch.equal(out_t, out_g), f"max |delta| {(out_t - out_g).abs().max().item()}"
    gin_t = _ext.ps_roipool_backward(gout, rois, H, W, TUNED)
    gin_g = _ext.ps_roipool_backward(gout, rois, H, W, GENERIC)
    torch.testing.assert_close(gin_t, gin_g, **TOL)
