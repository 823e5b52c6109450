"""GPU tests of the LAB kernels (lab/csrc/: strips 8 pixels wide, bf16x3, the round-2 16-wave strip kernel) -- correlation backward
kernels that lost their A/B measurements and left the product library in ABI 1.06.  They run only against the LAB build:

    make -C detect-to-track_amd/csrc lab
    D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/lib_lab/libd2t_ops.so python -m pytest tests/test_lab_kernels.py -m gpu

and are skipped with the product library (which rejects their selectors: tests/test_pointwise_correlation.py::test_empty_and_errors).
Same bars as the product's backward: <= 1e-5 of sum|terms| against the double-accumulating yardstick, the live reference, non-finite patterns.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _n(x):
    return x.detach().cpu().numpy()


@pytest.fixture(autouse=True)
def _lab_build_only():
    from detect_to_track.models import _native
    if not _native.IS_LAB_BUILD:
        pytest.skip("needs the lab build of libd2t_ops.so (make -C detect-to-track_amd/csrc lab; D2T_OPS_LIBRARY=.../lib_lab/libd2t_ops.so)")


@pytest.mark.parametrize("case", [(8, 256, 38, 63), (3, 20, 38, 75), (2, 300, 21, 44), (1, 64, 17, 130), (2, 40, 40, 20)], ids=str)
def test_bf16x3_backward(case, oracle):
    """D2T_IMPL_BF16X3 (opt-in): the backward on the bf16 matrix pipe, every f32 operand split into three bf16 pieces, six
    piece products per product (d2t_corr_bwd8bf.hip).  Held to the contract of the other gradients -- 1e-5 of the sum of
    |terms| of an element, against the yardstick with the reference's f32 terms added in double -- deterministic, and
    with signed data (cancellation: the error is relative to the terms, not to the result)."""
    from detect_to_track.models import _ext
    B, C, H, W = case
    rng = np.random.default_rng(B * 1000 + C)
    fm0, fm1 = rng.standard_normal((B, C, H, W)).astype(np.float32), rng.standard_normal((B, C, H, W)).astype(np.float32)
    gout = rng.standard_normal((B, H, W, 17, 17)).astype(np.float32)
    g0, g1 = _ext.pointwise_correlation_backward(_t(gout), _t(fm0), _t(fm1), 8, 1, 4)
    (w0, w1), (m0, m1) = oracle.corr_bwd_acc64(gout, fm0, fm1, 8, 1)
    oracle.assert_within_contract(_n(g0), w0, m0, 1e-5, "gradFM0, bf16x3")
    oracle.assert_within_contract(_n(g1), w1, m1, 1e-5, "gradFM1, bf16x3")
    h0, h1 = _ext.pointwise_correlation_backward(_t(gout), _t(fm0), _t(fm1), 8, 1, 4)
    assert torch.equal(g0, h0) and torch.equal(g1, h1)


@pytest.mark.parametrize("case", [(8, 256, 38, 63), (2, 512, 38, 75), (1, 260, 21, 30)], ids=str)
def test_bf16x3_backward_matches_live_reference(case, ref_modules):
    """The opt-in bf16x3 backward against the reference's own kernels on the same GPU, same inputs, at the tolerance the
    default backward is held to (the headline shape included)."""
    from detect_to_track.models import _ext
    ref_corr = ref_modules[0]
    B, C, H, W = case
    torch.manual_seed(4321)
    fm0, fm1 = torch.rand(B, C, H, W, device=DEV), torch.rand(B, C, H, W, device=DEV)
    gout = torch.rand(B, H, W, 17, 17, device=DEV)
    g0, g1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, 4)
    r0, r1 = ref_corr.pointwise_correlation_backward(gout, fm0, fm1, 8, 1)
    torch.testing.assert_close(g0, r0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g1, r1, rtol=1e-5, atol=1e-5)


def test_bf16x3_backward_nonfinite_and_huge_inputs(oracle):
    """A piece of Inf / NaN is NaN and a finite value above the bf16 range rounds to Inf: either way the tile's
    accumulators turn non-finite and the wave recomputes its region in the reference's form -- the non-finite pattern and
    every finite value are those of the f32 kernels."""
    from detect_to_track.models import _ext
    B, C, H, W = 2, 40, 21, 29
    torch.manual_seed(5)
    fm0, fm1 = torch.rand(B, C, H, W, device=DEV), torch.rand(B, C, H, W, device=DEV)
    gout = torch.rand(B, H, W, 17, 17, device=DEV)
    fm1[0, 3, 5, 7] = float("inf"); fm0[1, 0, 9, 20] = float("nan"); gout[0, 6, 6, 3, 4] = float("-inf")
    fm1[1, 7, 2, 3] = 3.4e38; fm0[0, 1, 1, 1] = -3.4e38; gout[1, 2, 2, 8, 8] = 1e-30
    g0, g1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, 4)
    r0, r1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, 1)
    for got, want in ((g0, r0), (g1, r1)):
        assert torch.equal(torch.isnan(got), torch.isnan(want))
        assert torch.equal(torch.isposinf(got), torch.isposinf(want)) and torch.equal(torch.isneginf(got), torch.isneginf(want))
        fin = torch.isfinite(want)
        torch.testing.assert_close(got[fin], want[fin], rtol=2e-5, atol=2e-3)   # elements next to the 3.4e38 values are ~1e37; the rest ~50


WIDE8, STRIP4 = 6, 7        # lab/csrc/d2t_lab_selectors.h: the two 8-wave backward kernels, demanded


@pytest.mark.parametrize("impl", [WIDE8, STRIP4], ids=["wide8", "strip4"])
@pytest.mark.parametrize("case", [(1, 32, 17, 24), (2, 130, 21, 30), (1, 128, 38, 63), (3, 40, 19, 75), (1, 16, 40, 21),
                                  (2, 200, 23, 64), (1, 300, 38, 20), (1, 7, 17, 25)], ids=str)
def test_strip_backward_kernels_match_oracle(case, impl, oracle):
    """Both 8-wave backward kernels, each demanded (no dispatch by grid size): strips 8 pixels wide x 128 channels
    (d2t_corr_bwd8w.hip: the window is not clamped to the map -- widths that are no multiple of 8 or narrower than one
    strip window of 24 columns, channel counts that are no multiple of 128, heights that are no multiple of 4) and strips 4 pixels wide
    (d2t_corr_bwd8.hip).  Signed data against the yardstick with the reference's f32 terms added in double
    (pointwise_correlation_cuda.cu:154-171): 1e-5 of the sum of |terms|; deterministic."""
    from detect_to_track.models import _ext
    B, C, H, W = case
    rng = np.random.default_rng(B * 1000 + C + W)
    fm0, fm1 = rng.standard_normal((B, C, H, W)).astype(np.float32), rng.standard_normal((B, C, H, W)).astype(np.float32)
    gout = rng.standard_normal((B, H, W, 17, 17)).astype(np.float32)
    g0, g1 = _ext.pointwise_correlation_backward(_t(gout), _t(fm0), _t(fm1), 8, 1, impl)
    (w0, w1), (m0, m1) = oracle.corr_bwd_acc64(gout, fm0, fm1, 8, 1)
    oracle.assert_within_contract(_n(g0), w0, m0, 1e-5, "gradFM0")
    oracle.assert_within_contract(_n(g1), w1, m1, 1e-5, "gradFM1")
    h0, h1 = _ext.pointwise_correlation_backward(_t(gout), _t(fm0), _t(fm1), 8, 1, impl)
    assert torch.equal(g0, h0) and torch.equal(g1, h1)


@pytest.mark.parametrize("impl", [WIDE8, STRIP4], ids=["wide8", "strip4"])
@pytest.mark.parametrize("case", [(8, 256, 38, 63), (2, 512, 38, 75), (1, 260, 21, 30)], ids=str)
def test_strip_backward_kernels_match_live_reference(case, impl, ref_modules):
    """Either kernel against the reference's own kernels on the same GPU, same inputs (the headline shape included)."""
    from detect_to_track.models import _ext
    ref_corr = ref_modules[0]
    B, C, H, W = case
    torch.manual_seed(4321)
    fm0, fm1 = torch.rand(B, C, H, W, device=DEV), torch.rand(B, C, H, W, device=DEV)
    gout = torch.rand(B, H, W, 17, 17, device=DEV)
    g0, g1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, impl)
    r0, r1 = ref_corr.pointwise_correlation_backward(gout, fm0, fm1, 8, 1)
    torch.testing.assert_close(g0, r0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g1, r1, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("impl", [WIDE8, STRIP4], ids=["wide8", "strip4"])
def test_strip_backward_kernels_nonfinite(impl):
    """Inf / NaN in the maps and in gradOut -- also right at the map's left and right borders, where the unclamped window of
    the 8-pixel kernel multiplies slots outside the map by G = 0: pattern and finite values of the reference-order kernels."""
    from detect_to_track.models import _ext
    B, C, H, W = 2, 40, 21, 29
    torch.manual_seed(5)
    fm0, fm1 = torch.rand(B, C, H, W, device=DEV), torch.rand(B, C, H, W, device=DEV)
    gout = torch.rand(B, H, W, 17, 17, device=DEV)
    fm1[0, 3, 5, 7] = float("inf"); fm0[1, 0, 9, 20] = float("nan"); gout[0, 6, 6, 3, 4] = float("-inf")
    fm1[1, 9, 4, W - 1] = float("nan"); fm0[0, 2, 11, 0] = float("inf"); fm1[0, 0, 0, 0] = float("nan"); fm0[1, C - 1, H - 1, W - 1] = float("-inf")
    gout[1, 7, 9, 16, 16] = float("nan")          # a cell the reference never reads
    g0, g1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, impl)
    r0, r1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, 1)
    for got, want in ((g0, r0), (g1, r1)):
        assert torch.equal(torch.isnan(got), torch.isnan(want))
        assert torch.equal(torch.isposinf(got), torch.isposinf(want)) and torch.equal(torch.isneginf(got), torch.isneginf(want))
        fin = torch.isfinite(want)
        torch.testing.assert_close(got[fin], want[fin], rtol=1e-5, atol=1e-5)
