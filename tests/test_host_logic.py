"""CPU: the Python surface mirrors the reference's (names, constructor arguments, attributes,
error classes, backward arities) and the autograd wiring is right.

There is no CPU implementation of the ops (as in the reference: common/cpp_common.hpp:1), so the
wiring test substitutes the CPU ORACLE for the native entry points -- inside this test only --
and runs the reference's own gradcheck parametrisations in float64 on CPU tensors.
"""
import numpy as np
import pytest
import torch
from torch.autograd import gradcheck

import detect_to_track.models as models
from detect_to_track.models import (PSROIPool, PSROIPoolFunction, PointwiseCorrelation,
                                    PointwiseCorrelationFunction, ROIPool, ROIPoolFunction, _ext)


def test_exports_match_reference_init():
    # reference models/__init__.py:3-5
    from detect_to_track.models.ps_roipool.ps_roipool import PSROIPool as A
    from detect_to_track.models.pointwise_correlation.pointwise_correlation import PointwiseCorrelation as B
    from detect_to_track.models.roipool.roipool import ROIPool as C
    assert (A, B, C) == (models.PSROIPool, models.PointwiseCorrelation, models.ROIPool)


def test_modules_hold_only_their_hyperparameters():
    pc, rp, ps = PointwiseCorrelation(8, 1), ROIPool(7), PSROIPool(31, 7)
    assert (pc.d_max, pc.stride, rp.r_hw, ps.n_targets, ps.r_hw) == (8, 1, 7, 31, 7)
    for m in (pc, rp, ps):
        assert list(m.parameters()) == [] and list(m.buffers()) == [] and m.state_dict() == {}
        m.cuda if False else None


def test_cpu_tensors_are_rejected_like_the_reference():
    with pytest.raises(RuntimeError, match="CPU op not implemented"):
        PointwiseCorrelation(3, 1)(torch.rand(1, 2, 6, 6), torch.rand(1, 2, 6, 6))
    with pytest.raises(RuntimeError, match="CPU op not implemented"):
        ROIPool(3)(torch.rand(2, 6, 6), torch.rand(1, 4))
    with pytest.raises(RuntimeError, match="CPU op not implemented"):
        PSROIPool(2, 3)(torch.rand(18, 6, 6), torch.rand(1, 4))


def test_psroipool_channel_mismatch_is_a_value_error():
    # raised before native code is reached (reference ps_roipool.py:44-49)
    with pytest.raises(ValueError, match="18"):
        PSROIPool(2, 3)(torch.rand(17, 6, 6), torch.rand(1, 4))


@pytest.fixture
def oracle_backend(monkeypatch, oracle):
    """Route the six _ext entry points to the oracle for CPU float64 tensors (test-only)."""
    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def n(x):
        return x.detach().contiguous().numpy()

    monkeypatch.setattr(_ext, "pointwise_correlation_forward",
                        lambda a, b, d, s: t(oracle.corr_fwd(n(a), n(b), d, s)))
    monkeypatch.setattr(_ext, "pointwise_correlation_backward",
                        lambda g, a, b, d, s: tuple(t(x) for x in oracle.corr_bwd(n(g), n(a), n(b), d, s)))
    monkeypatch.setattr(_ext, "roipool_forward", lambda f, r, k: t(oracle.roipool_fwd(n(f), n(r), k)))
    monkeypatch.setattr(_ext, "roipool_backward", lambda g, r, h, w: t(oracle.roipool_bwd(n(g), n(r), h, w)))
    monkeypatch.setattr(_ext, "ps_roipool_forward", lambda f, r, nt, k: t(oracle.psroipool_fwd(n(f), n(r), nt, k)))
    monkeypatch.setattr(_ext, "ps_roipool_backward", lambda g, r, h, w: t(oracle.psroipool_bwd(n(g), n(r), h, w)))


@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("input_b", [1, 2])
@pytest.mark.parametrize("input_hw", [10, 11])
def test_correlation_autograd_wiring(stride, input_b, input_hw, oracle_backend):
    pc = PointwiseCorrelation(3, stride)
    fm0 = torch.rand(input_b, 2, input_hw, input_hw, dtype=torch.float64, requires_grad=True)
    fm1 = torch.rand(input_b, 2, input_hw, input_hw, dtype=torch.float64, requires_grad=True)
    assert gradcheck(pc, (fm0, fm1))
    out = PointwiseCorrelationFunction.apply(fm0, fm1, 3, stride)
    assert out.shape == (input_b, input_hw, input_hw, 7, 7)
    # non-contiguous incoming gradient is made contiguous (reference pointwise_correlation.py:62)
    out.backward(torch.rand(7, 7, input_b, input_hw, input_hw, dtype=torch.float64).permute(2, 3, 4, 0, 1))
    assert fm0.grad.shape == fm0.shape and fm1.grad.shape == fm1.shape


@pytest.mark.parametrize("r_hw", [5, 6])
@pytest.mark.parametrize("fm_h,fm_w", [(10, 10), (10, 11), (11, 10), (11, 11)])
def test_roipool_autograd_wiring(r_hw, fm_h, fm_w, oracle_backend):
    fm = torch.rand(2, fm_h, fm_w, dtype=torch.float64, requires_grad=True)
    rois = torch.tensor([[0.5, 0.5, 0.5, 0.5], [0.1, 0.1, 0.2, 0.3]], dtype=torch.float64)
    assert gradcheck(ROIPool(r_hw), (fm, rois))
    rois_g = rois.clone().requires_grad_(True)
    ROIPoolFunction.apply(fm, rois_g, r_hw).sum().backward()
    assert rois_g.grad is None                      # no gradient to the boxes (reference roipool.py:57)


@pytest.mark.parametrize("n_targets", [1, 2])
@pytest.mark.parametrize("r_hw", [6, 7])
def test_psroipool_autograd_wiring(n_targets, r_hw, oracle_backend):
    fm = torch.rand(n_targets * r_hw ** 2, 10, 11, dtype=torch.float64, requires_grad=True)
    rois = torch.tensor([[0.5, 0.5, 0.1, 0.1], [0.1, 0.1, 0.2, 0.3], [1.5, 1.5, 0.2, 0.2]], dtype=torch.float64)
    assert gradcheck(PSROIPool(n_targets, r_hw), (fm, rois))
    out = PSROIPoolFunction.apply(fm, rois, n_targets, r_hw)
    assert out.shape == (3, n_targets, r_hw, r_hw)


def test_oracle_known_answers(oracle):
    """Facts recorded from an execution of the reference kernel bodies."""
    o = oracle.corr_fwd(np.ones((1, 2, 10, 10)), np.ones((1, 2, 10, 10)), 3, 1)
    blk = o[0, 5, 5]
    assert (blk[:6, :6] == 2).all() and not blk[6].any() and not blk[:, 6].any()        # F3
    o2 = oracle.corr_fwd(np.ones((1, 2, 10, 10)), np.ones((1, 2, 10, 10)), 3, 2)
    assert sorted(set(np.nonzero(o2[0, 0, 0])[0])) == [3, 5]                              # Appendix A.1
    assert sorted(set(np.nonzero(o2[0, 5, 5])[0])) == [0, 2, 4]
    assert len(set(oracle.psroipool_channels(2, 3).ravel().tolist())) == 13              # F5
    nan = oracle.roipool_fwd(np.ones((1, 10, 10), np.float32), np.asarray([[.5, .5, 0, 0]], np.float32), 2)
    assert np.isnan(nan).all()                                                           # F6
    z = oracle.psroipool_fwd(np.full((72, 10, 11), 10.0, np.float32), np.asarray([[3, 3, .5, .5]], np.float32), 2, 6)
    assert not z.any()                                                                   # tests/test_ps_roipool.py:44
