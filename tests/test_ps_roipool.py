"""GPU parity tests of PSROIPool (position-sensitive average pooling) through the C ABI.

Mirrors reference tests/test_ps_roipool.py:8-44 (gradcheck f64 incl. an out-of-bounds RoI, and
the known-answer out-of-bounds -> zeros test) and adds absolute parity against the CPU oracle,
the reference-generated fixtures and the reference's kernels live.

Bit-exact: cell bounds, the (t+1)*(i*k+j) channel map, forward values (same running-sum order
as the reference).  Backward: |delta| <= 1e-5 (atomics in the reference, order undefined).
"""
import numpy as np
import pytest
import torch
from torch.autograd import gradcheck

from conftest import ADVERSARIAL_ROIS, golden_files, golden_ids, load_golden, random_rois

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL32 = dict(rtol=1e-5, atol=1e-5)
TOL64 = dict(rtol=1e-12, atol=1e-12)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _n(x):
    return x.detach().cpu().numpy()


@pytest.mark.parametrize("n_targets", [1, 2])
@pytest.mark.parametrize("r_hw", [6, 7])
@pytest.mark.parametrize("fm_h", [10, 11])
@pytest.mark.parametrize("fm_w", [10, 11])
def test_ps_roipool_gradients(n_targets, r_hw, fm_h, fm_w):
    from detect_to_track.models import PSROIPool
    pr = PSROIPool(n_targets, r_hw)
    fm = torch.rand(n_targets * r_hw ** 2, fm_h, fm_w).double().cuda().requires_grad_(True)
    rois = (torch.Tensor([[0.5, 0.5, 0.1, 0.1], [0.1, 0.1, 0.2, 0.3], [1.5, 1.5, 0.2, 0.2]])
            .double().cuda().requires_grad_(False))
    assert gradcheck(pr, (fm, rois))


@pytest.mark.parametrize("n_targets", [1, 2])
@pytest.mark.parametrize("r_hw", [6, 7])
@pytest.mark.parametrize("fm_h", [10, 11])
@pytest.mark.parametrize("fm_w", [10, 11])
def test_ps_roipool_can_handle_oob(n_targets, r_hw, fm_h, fm_w):
    # torch.full(shape, 10) is int64 on modern torch (float on the reference's torch 1.1): use 10.0
    from detect_to_track.models import PSROIPool
    pr = PSROIPool(n_targets, r_hw).cuda()
    fm = torch.full((n_targets * r_hw ** 2, fm_h, fm_w), 10.0).cuda()
    rois = torch.as_tensor([[3.0, 3.0, 0.5, 0.5]]).cuda()
    ans = pr(fm, rois).cpu()
    assert torch.allclose(ans, torch.zeros(len(rois), n_targets, r_hw, r_hw))


def _check_bounds(bounds, ref_bounds):
    empty = (ref_bounds[..., 0] < 0)
    ours_empty = (bounds[..., 1] <= bounds[..., 0]) | (bounds[..., 3] <= bounds[..., 2])
    np.testing.assert_array_equal(ours_empty, empty)
    np.testing.assert_array_equal(bounds[~empty], ref_bounds[~empty])


@pytest.mark.parametrize("path", [p for p in golden_files("psroipool") if "known_answer" not in p.stem],
                         ids=[i for i in golden_ids("psroipool") if "known_answer" not in i])
def test_matches_reference_fixture(path):
    from detect_to_track.models import _ext
    g = load_golden(path)
    nT, k = int(g["nT"]), int(g["k"])
    _, H, W = g["fm"].shape
    tol = TOL32 if g["fm"].dtype == np.float32 else TOL64
    out = _n(_ext.ps_roipool_forward(_t(g["fm"]), _t(g["rois"]), nT, k))
    np.testing.assert_array_equal(out, g["out"])                               # bit-exact forward
    gin = _n(_ext.ps_roipool_backward(_t(g["gout"]), _t(g["rois"]), H, W))
    np.testing.assert_allclose(gin, g["gin"], **tol)
    _check_bounds(_n(_ext.roipool_bins(_t(g["rois"]), H, W, k, position_sensitive=True)), g["bounds"])
    np.testing.assert_array_equal(_n(_ext.ps_roipool_channels(nT, k, DEV)), g["channels"])


def test_known_answer_fixture():
    from detect_to_track.models import _ext
    g = load_golden([p for p in golden_files("psroipool") if "known_answer" in p.stem][0])
    fm = torch.full((2 * 49, 10, 11), 10.0, device=DEV)
    out = _ext.ps_roipool_forward(fm, torch.tensor([[3.0, 3.0, 0.5, 0.5]], device=DEV), 2, 7)
    np.testing.assert_array_equal(_n(out), g["out"])
    assert not g["out"].any()


CASES = [  # (R, nT, H, W, k)
    (3, 1, 10, 10, 6), (11, 3, 38, 63, 7), (11, 5, 9, 14, 3), (24, 4, 38, 75, 7), (300, 21, 38, 63, 7),
    (64, 31, 38, 75, 7), (5, 2, 11, 10, 7), (1, 1, 3, 3, 1), (130, 4, 20, 20, 5),
    (1200, 13, 38, 75, 7), (1000, 16, 21, 30, 7),        # backward as a GEMM (auto dispatch from 12 targets up)
    (260, 32, 9, 100, 7), (1100, 9, 38, 63, 7), (40, 33, 12, 20, 7),   # 32 targets x 7 column tiles; 8..11 targets at R >= 1000; 33 targets: not the GEMM
    # the sorted-corner-list backward (d2t_pool_sorted.hip) under DEFAULT dispatch: the GEMM form does not apply (more than
    # 32 targets / maps wider than 128 columns) and there are >= 12 targets with R * nT >= 14000
    (500, 33, 38, 63, 7), (800, 20, 20, 140, 7),
]


@pytest.mark.parametrize("impl", [0, 1], ids=["auto", "generic"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", CASES, ids=str)
def test_matches_oracle(case, dtype, impl, oracle):
    from detect_to_track.models import _ext
    R, nT, H, W, k = case
    rng = np.random.default_rng(hash(case) % 2**32)
    rois = (np.asarray(ADVERSARIAL_ROIS, dtype=dtype) if R == 11 else random_rois(R, R + nT, dtype))
    fm = rng.random((nT * k * k, H, W)).astype(dtype)
    gout = rng.random((R, nT, k, k)).astype(dtype)
    tol = TOL32 if dtype == np.float32 else TOL64
    out = _n(_ext.ps_roipool_forward(_t(fm), _t(rois), nT, k, impl))
    np.testing.assert_array_equal(out, oracle.psroipool_fwd(fm, rois, nT, k))
    gin = _n(_ext.ps_roipool_backward(_t(gout), _t(rois), H, W, impl))
    np.testing.assert_allclose(gin, oracle.psroipool_bwd(gout, rois, H, W), **tol)
    np.testing.assert_array_equal(_n(_ext.roipool_bins(_t(rois), H, W, k, position_sensitive=True)),
                                  oracle.roipool_bins(rois, H, W, k, position_sensitive=True))
    np.testing.assert_array_equal(_n(_ext.ps_roipool_channels(nT, k, DEV)), oracle.psroipool_channels(nT, k))


@pytest.mark.parametrize("case", [(300, 21, 38, 63, 7), (3000, 31, 38, 75, 7), (3000, 4, 38, 75, 7), (11, 3, 38, 63, 7),
                                  (500, 33, 38, 63, 7), (800, 20, 20, 140, 7)],     # the last two: sorted corner lists
                         ids=str)
def test_matches_live_reference(case, ref_modules, oracle):
    from detect_to_track.models import _ext
    ref_ps = ref_modules[2]
    R, nT, H, W, k = case
    torch.manual_seed(7)
    fm = torch.rand(nT * k * k, H, W, device=DEV)
    rois = _t(np.asarray(ADVERSARIAL_ROIS, np.float32) if R == 11 else random_rois(R, 1))
    gout = torch.rand(R, nT, k, k, device=DEV)
    out = _ext.ps_roipool_forward(fm, rois, nT, k)
    assert torch.equal(out, ref_ps.ps_roipool_forward(fm, rois, nT, k))
    gin = _ext.ps_roipool_backward(gout, rois, H, W)
    # The reference adds f32 terms with atomicAdd in an undefined order (ps_roipool_cuda.cu:137); channel 0 collects
    # R*nT of them per pixel.  The contract is 1e-5 of an element's magnitude scale sum|terms|: the kernel against the
    # exact sum of the reference's own f32 terms (oracle, f64 accumulator), and against the reference's kernel itself.
    want, mag = oracle.psroipool_bwd_acc64(_n(gout), _n(rois), H, W)
    oracle.assert_within_contract(_n(gin), want, mag, 1e-5, "PSROIPool gradient vs exact sum of the reference's terms")
    oracle.assert_within_contract(_n(gin), _n(ref_ps.ps_roipool_backward(gout, rois, H, W)), mag, 1e-5,
                                  "PSROIPool gradient vs the reference's atomics")
    assert torch.equal(gin, _ext.ps_roipool_backward(gout, rois, H, W))        # deterministic


def test_sorted_lists_backward_adversarial_rois(oracle):
    """The many-target backward designs (the GEMM form from 12 targets up; the sorted corner lists under
    D2T_PS_BWD=sorted) on the adversarial RoIs -- clamped, empty, reversed and one-pixel bins -- repeated among random ones,
    so that equal corner addresses span many list steps."""
    from detect_to_track.models import _ext
    nT, H, W, k = 16, 38, 63, 7
    adv = np.asarray(ADVERSARIAL_ROIS, np.float32)
    rois = np.concatenate([adv] * 60 + [random_rois(500, 3)], 0)
    rois = rois[np.random.default_rng(5).permutation(len(rois))]
    gout = np.random.default_rng(6).standard_normal((len(rois), nT, k, k)).astype(np.float32)
    gin = _n(_ext.ps_roipool_backward(_t(gout), _t(rois), H, W))
    # up to ~16 * 1160 terms of magnitude ~1 per pixel of channel 0: the yardstick keeps the reference's f32 geometry
    # and f32 terms g / n (an f64 oracle would move floor/ceil bin bounds of these RoIs) and adds them in double;
    # the kernel is held to 1e-5 of sum|terms| per element.
    want, mag = oracle.psroipool_bwd_acc64(gout, rois, H, W)
    oracle.assert_within_contract(gin, want, mag, 1e-5, "PSROIPool gradient")
    assert np.array_equal(gin, _n(_ext.ps_roipool_backward(_t(gout), _t(rois), H, W)))


def test_bin_row_forward_adversarial_and_nonfinite_rois(oracle):
    """The bin-row forward (d2t_pool_tuned.hip: RoIs ordered by cell size, seven channels in LDS) on the adversarial RoIs --
    clamped, empty, reversed and one-pixel bins -- plus RoIs with huge, negative, infinite and NaN extents (their order
    bucket is arbitrary, their cells are the reference's), bit for bit; and a map with non-finite pixels: a masked lane
    reads pixels outside its cell and must drop them."""
    from detect_to_track.models import _ext
    nT, H, W, k = 16, 38, 75, 7
    adv = np.asarray(ADVERSARIAL_ROIS, np.float32)
    odd = np.array([[0.5, 0.5, 40.0, 40.0], [0.5, 0.5, -0.3, 0.4], [0.2, 0.8, np.inf, 0.1], [0.5, 0.5, np.nan, 0.2],
                    [np.nan, 0.5, 0.2, 0.2], [0.3, 0.3, 0.2, -np.inf], [1e30, -1e30, 1e30, 1e30]], np.float32)
    rois = np.concatenate([adv] * 20 + [odd] * 8 + [random_rois(400, 4)], 0)
    rois = rois[np.random.default_rng(7).permutation(len(rois))]
    assert len(rois) * nT >= 6000                                              # the bin-row form's dispatch threshold
    fm = np.random.default_rng(8).standard_normal((nT * k * k, H, W)).astype(np.float32)
    np.testing.assert_array_equal(_n(_ext.ps_roipool_forward(_t(fm), _t(rois), nT, k)), oracle.psroipool_fwd(fm, rois, nT, k))
    fm[::5, ::7, ::3] = np.inf
    fm[1::5, 3::7, 1::3] = np.nan
    got, want = _n(_ext.ps_roipool_forward(_t(fm), _t(rois), nT, k)), oracle.psroipool_fwd(fm, rois, nT, k)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(np.nan_to_num(got, nan=0.0), np.nan_to_num(want, nan=0.0))


@pytest.mark.parametrize("case", [(5000, 2, 20, 30), (400, 16, 60, 100), (4096, 2, 38, 75), (1000, 7, 1, 1), (700, 9, 3, 300)], ids=str)
def test_forward_dispatch_edges(case, oracle):
    """More than 4096 RoIs and maps whose seven channels do not fit in LDS take the channel-resident kernel; exactly
    4096 RoIs, a one-pixel map and a 3 x 300 map the bin-row form.  All bit-identical to the oracle."""
    from detect_to_track.models import _ext
    R, nT, H, W = case
    rng = np.random.default_rng(R + nT)
    rois = random_rois(R, R + 1)
    fm = rng.random((nT * 49, H, W)).astype(np.float32)
    np.testing.assert_array_equal(_n(_ext.ps_roipool_forward(_t(fm), _t(rois), nT, 7)), oracle.psroipool_fwd(fm, rois, nT, 7))


@pytest.mark.parametrize("case", [(400, 16, 38, 63), (1200, 9, 20, 30)], ids=str)
def test_nonfinite_gradout_gemm_backward(case):
    """Inf / NaN in gradOut on the shapes that take the backward GEMM: 0 x Inf = NaN must not reach columns outside the
    value's cell (the reference, ps_roipool_cuda.cu:131-139, only adds to the cell's own pixels): a poisoned task is
    recomputed with exact membership.  Pattern and finite values as the type-generic kernel."""
    from detect_to_track.models import _ext
    R, nT, H, W = case
    rng = np.random.default_rng(R + nT)
    rois = _t(random_rois(R, 9))
    gout = rng.standard_normal((R, nT, 7, 7)).astype(np.float32)
    gout[3, 5, 2, 2] = np.inf; gout[R - 1, nT - 1, 6, 0] = -np.inf; gout[R // 2, 1, 0, 6] = np.nan
    got = _ext.ps_roipool_backward(_t(gout), rois, H, W)
    want = _ext.ps_roipool_backward(_t(gout), rois, H, W, 1)
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    assert torch.equal(torch.isposinf(got), torch.isposinf(want)) and torch.equal(torch.isneginf(got), torch.isneginf(want))
    fin = torch.isfinite(want)
    torch.testing.assert_close(got[fin], want[fin], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("case", [(1500, 4, 38, 75), (3000, 16, 38, 63), (1401, 7, 19, 40), (4500, 2, 38, 75)], ids=str)
def test_row_form_backward_roi_ranges(case, oracle):
    """The row-form backward (k_ps_bwd_rows, round 5) deals a task's RoIs to TWO workgroups from 1,400 RoIs up (<= 16 targets) and the
    gather adds the partial planes; 4,500 RoIs in two ranges also take a second round of the hit scan (1,792 RoIs per round).  Signed data against the
    yardstick with the reference's f32 terms added in double (ps_roipool_cuda.cu:120-139): 1e-5 of the sum of |terms|; Inf / NaN in
    gradOut: the pattern and the finite values of the reference-order kernels; deterministic."""
    from detect_to_track.models import _ext
    R, nT, H, W = case
    rng = np.random.default_rng(R + nT)
    rois = random_rois(R, 5)
    gout = rng.standard_normal((R, nT, 7, 7)).astype(np.float32)
    got = _ext.ps_roipool_backward(_t(gout), _t(rois), H, W)
    want, mass = oracle.psroipool_bwd_acc64(gout, rois, H, W)
    oracle.assert_within_contract(_n(got), want, mass, 1e-5, "gradFM, row form with RoI ranges")
    assert torch.equal(got, _ext.ps_roipool_backward(_t(gout), _t(rois), H, W))
    gout[3, 1, 2, 2] = np.inf; gout[R - 1, nT - 1, 6, 0] = -np.inf; gout[R // 2 + 1, 0, 0, 6] = np.nan
    got = _ext.ps_roipool_backward(_t(gout), _t(rois), H, W)
    ref = _ext.ps_roipool_backward(_t(gout), _t(rois), H, W, 1)
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    assert torch.equal(torch.isposinf(got), torch.isposinf(ref)) and torch.equal(torch.isneginf(got), torch.isneginf(ref))
    fin = torch.isfinite(ref)
    torch.testing.assert_close(got[fin], ref[fin], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", [(300, 21, 38, 63, 6, "random"), (40, 31, 20, 31, 3, "random"), (17, 5, 9, 14, 14, "random"), (9, 3, 7, 5, 1, "random"),
                                  (300, 4, 38, 63, 8, "adversarial"), (520, 9, 12, 17, 2, "random"), (12, 6, 20, 20, 3, "huge")], ids=str)
def test_outside_the_envelope_backward_equals_generic_kernel(case, dtype):
    """Cell counts other than 7 (and all of f64) take the per-(cell, pixel) RoI lists of d2t_pool_lists.hip under the default dispatch:
    the same terms gradOut / n in the same order (bins ascending, RoIs ascending, targets ascending) as the thread-per-pixel kernel
    (ps_roipool_cuda.cu:118-127, gather form): bit-identical to D2T_IMPL_GENERIC, which test_matches_oracle pins to the oracle.
    "huge": RoIs many times the map, whose lists exceed the workspace -- the device-side fallback.  Non-finite gradOut included."""
    from detect_to_track.models import _ext
    R, nT, H, W, k, kind = case
    rng = np.random.default_rng(R * 31 + k)
    if kind == "random":
        rois = random_rois(R, k)
    elif kind == "adversarial":
        rois = np.resize(ADVERSARIAL_ROIS, (R, 4))
    else:
        rois = np.concatenate([rng.random((R, 2)), 20 + 30 * rng.random((R, 2))], axis=1)
    gout = rng.standard_normal((R, nT, k, k))
    gout[0, 0, 0, 0] = np.inf
    gout[R - 1, nT - 1, k - 1, k - 1] = np.nan
    g, r = _t(gout.astype(dtype)), _t(np.asarray(rois).astype(dtype))
    got = _ext.ps_roipool_backward(g, r, H, W, 0)
    want = _ext.ps_roipool_backward(g, r, H, W, 1)
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    fin = ~torch.isnan(want)
    assert torch.equal(got[fin], want[fin])


def test_channel_collisions_and_unused_channels():
    """(t+1)*(i*k+j) is many-to-one (reference ps_roipool_cuda.cu:58): for nT=2,k=3 only 13 of 18
    channels are ever read; the gradient of the other 5 must be exactly zero."""
    from detect_to_track.models import _ext
    nT, k, H, W = 2, 3, 12, 12
    ch = _n(_ext.ps_roipool_channels(nT, k, DEV))
    used = sorted(set(ch.ravel().tolist()))
    assert len(used) == 13
    gout = torch.ones(4, nT, k, k, device=DEV)
    gin = _ext.ps_roipool_backward(gout, _t(random_rois(4, 3)), H, W)
    unused = [c for c in range(nT * k * k) if c not in used]
    assert not gin[unused].any()
    assert gin[0].sum() > 0      # bin 0 of every target reads channel 0


def test_empty_and_errors():
    from detect_to_track.models import PSROIPool
    pr = PSROIPool(2, 3)
    out = pr(torch.rand(18, 10, 10, device=DEV), torch.empty(0, 4, device=DEV))
    assert out.shape == (0, 2, 3, 3)
    with pytest.raises(ValueError):
        pr(torch.rand(17, 10, 10, device=DEV), torch.rand(1, 4, device=DEV))
    with pytest.raises(RuntimeError, match="CPU op not implemented"):
        pr(torch.rand(18, 10, 10), torch.rand(1, 4))
