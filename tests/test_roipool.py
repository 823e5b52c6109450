"""GPU parity tests of ROIPool (average pooling), HIP kernels through the C ABI.

Mirrors reference tests/test_roipool.py:10-27 (gradcheck, f64) and adds absolute parity against
the CPU oracle, the reference-generated golden fixtures and the reference's kernels live.

Bit-exact: integer bin bounds {i0,i1,j0,j1}, the NaN pattern of empty bins, and the FORWARD values
of the type-generic kernel (the reference's row-major running sum and IEEE divide,
roipool_cuda.cu:56-61; it also serves every call with fewer than 32 RoIs under impl = auto).
The tuned forward (>= 32 RoIs, k <= 16) reads summed-area tables: the exactly rounded bin sum instead of the
reference's f32 running sum, held to |delta| <= 1e-5 abs/rel (BASELINE.json) with the NaN pattern
bit-exact.  Backward: |delta| <= 1e-5 -- the reference sums with atomics in undefined order.
"""
import numpy as np
import pytest
import torch
from torch.autograd import gradcheck

from conftest import ADVERSARIAL_ROIS, NEGATIVE_ROIS, golden_files, golden_ids, load_golden, random_rois

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL32 = dict(rtol=1e-5, atol=1e-5)
TOL64 = dict(rtol=1e-12, atol=1e-12)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _n(x):
    return x.detach().cpu().numpy()


@pytest.mark.parametrize("r_hw", [5, 6])
@pytest.mark.parametrize("fm_c", [2])
@pytest.mark.parametrize("fm_h", [10, 11])
@pytest.mark.parametrize("fm_w", [10, 11])
def test_roipool_gradients(r_hw, fm_c, fm_h, fm_w):
    from detect_to_track.models import ROIPool
    rp = ROIPool(r_hw)
    fm = torch.rand(fm_c, fm_h, fm_w).double().cuda().requires_grad_(True)
    rois = torch.Tensor([[0.5, 0.5, 0.5, 0.5], [0.1, 0.1, 0.2, 0.3]]).double().cuda().requires_grad_(False)
    assert gradcheck(rp, (fm, rois))


def _assert_fwd(out, want, exact):
    """exact: bit for bit (NaNs in the same places); else NaN pattern exact, values to 1e-5."""
    out, want = np.asarray(out), np.asarray(want)
    np.testing.assert_array_equal(np.isnan(out), np.isnan(want))
    if exact:
        np.testing.assert_array_equal(out, want)
    else:
        np.testing.assert_allclose(np.nan_to_num(out), np.nan_to_num(want), **TOL32)


def _check_bounds(bounds, ref_bounds):
    """ref_bounds has -1 rows for bins the reference leaves empty; ours must be empty there."""
    empty = (ref_bounds[..., 0] < 0)
    ours_empty = (bounds[..., 1] <= bounds[..., 0]) | (bounds[..., 3] <= bounds[..., 2])
    np.testing.assert_array_equal(ours_empty, empty)
    np.testing.assert_array_equal(bounds[~empty], ref_bounds[~empty])


@pytest.mark.parametrize("path", golden_files("roipool"), ids=golden_ids("roipool"))
def test_matches_reference_fixture(path):
    from detect_to_track.models import _ext
    g = load_golden(path)
    k = int(g["k"])
    C, H, W = g["fm"].shape
    tol = TOL32 if g["fm"].dtype == np.float32 else TOL64
    out = _n(_ext.roipool_forward(_t(g["fm"]), _t(g["rois"]), k))
    np.testing.assert_array_equal(out, g["out"])                          # bit-exact, NaNs included (k = 7: the direct kernel; else < 32 RoIs: generic)
    gin = _n(_ext.roipool_backward(_t(g["gout"]), _t(g["rois"]), H, W))
    np.testing.assert_allclose(gin, g["gin"], **tol)
    if g["fm"].dtype == np.float32:                                       # the tuned kernels, demanded
        # the tuned forward against what the REFERENCE's kernel produced: k = 7 runs the direct kernel (round 6) -- its bits; other k the summed-area tables
        _assert_fwd(_n(_ext.roipool_forward(_t(g["fm"]), _t(g["rois"]), k, 2)), g["out"], exact=(k == 7 and H * W <= 4800))
        np.testing.assert_allclose(_n(_ext.roipool_backward(_t(g["gout"]), _t(g["rois"]), H, W, 2)), g["gin"], **tol)
    _check_bounds(_n(_ext.roipool_bins(_t(g["rois"]), H, W, k)), g["bounds"])


CASES = [  # (R, C, H, W, k)
    (2, 2, 10, 10, 5), (11, 3, 38, 63, 7), (11, 2, 9, 14, 3), (24, 8, 38, 63, 7), (5, 64, 38, 75, 7),
    (40, 70, 20, 33, 7), (3, 1, 5, 5, 1), (16, 130, 38, 63, 7), (7, 256, 38, 63, 7), (9, 5, 38, 63, 2),
    (70, 9, 38, 75, 7), (33, 3, 100, 140, 7), (64, 5, 7, 9, 7),
    # backward as a GEMM (W <= 128): 8 column tiles; 300 RoIs on a 5-row map (every bin row of a RoI contains
    # the same map rows: up to 49 slots per RoI and row); channels not a multiple of the 32 of a task
    (37, 19, 21, 120, 7), (300, 33, 5, 40, 7), (270, 45, 38, 63, 7),
    # bin counts other than 7 with >= 32 RoIs: the summed-area forward with k at run time (k <= 16), bin lists in the backward
    (40, 6, 20, 33, 6), (64, 5, 38, 63, 3), (33, 4, 38, 63, 14), (50, 3, 30, 40, 16), (35, 2, 12, 12, 17), (300, 7, 38, 63, 1),
]


@pytest.mark.parametrize("impl", [0, 1, 2], ids=["auto", "generic", "tuned"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", CASES, ids=str)
def test_matches_oracle(case, dtype, impl, oracle):
    from detect_to_track.models import _ext
    R, C, H, W, k = case
    rng = np.random.default_rng(hash(case) % 2**32)
    rois = (np.asarray(ADVERSARIAL_ROIS, dtype=dtype) if R == 11 else random_rois(R, R * 7 + C, dtype))
    fm = rng.random((C, H, W)).astype(dtype)
    gout = rng.random((R, C, k, k)).astype(dtype)
    tol = TOL32 if dtype == np.float32 else TOL64
    if impl == 2 and (dtype == np.float64 or k > 16):
        pytest.skip("the tuned kernels are f32 (forward: k <= 16; backward: k = 7, else the bin lists)")
    out = _n(_ext.roipool_forward(_t(fm), _t(rois), k, impl))
    # round 6: k = 7 runs d2t_roipool_fwd_direct.hip (the reference's order: bit-identical); the summed-area tables (within 1e-5) serve k != 7
    direct = k == 7 and H * W * 32 + 4800 <= 160 * 1024           # (maps whose 8 interleaved planes fit the LDS)
    sat_fwd = dtype == np.float32 and k <= 16 and not direct and (impl == 2 or (impl == 0 and R >= 32))
    _assert_fwd(out, oracle.roipool_fwd(fm, rois, k), exact=not sat_fwd)
    gin = _n(_ext.roipool_backward(_t(gout), _t(rois), H, W, impl))
    np.testing.assert_allclose(gin, oracle.roipool_bwd(gout, rois, H, W), **tol)
    np.testing.assert_array_equal(_n(_ext.roipool_bins(_t(rois), H, W, k)), oracle.roipool_bins(rois, H, W, k))


@pytest.mark.parametrize("case", [(300, 1024, 38, 63, 7), (37, 1891, 38, 75, 7), (11, 6, 38, 63, 7)], ids=str)
def test_matches_live_reference(case, ref_modules):
    from detect_to_track.models import _ext
    ref_roi = ref_modules[1]
    R, C, H, W, k = case
    torch.manual_seed(99)
    fm = torch.rand(C, H, W, device=DEV)
    rois = _t(np.asarray(ADVERSARIAL_ROIS, np.float32) if R == 11 else random_rois(R, 0))
    gout = torch.rand(R, C, k, k, device=DEV)
    out = _ext.roipool_forward(fm, rois, k)
    ref = ref_roi.roipool_forward(fm, rois, k)
    _assert_fwd(_n(out), _n(ref), exact=True)                             # round 6: the tuned k = 7 forward is the reference's arithmetic, bit for bit
    _assert_fwd(_n(_ext.roipool_forward(fm, rois, k, 2)), _n(ref), exact=True)
    gen = _ext.roipool_forward(fm, rois, k, 1)                            # the generic kernel: bit-exact at any size
    _assert_fwd(_n(gen), _n(ref), exact=True)
    gin = _ext.roipool_backward(gout, rois, H, W)
    torch.testing.assert_close(gin, ref_roi.roipool_backward(gout, rois, H, W), **TOL32)


@pytest.mark.parametrize("impl", [0, 1, 2], ids=["auto", "generic", "tuned"])
@pytest.mark.parametrize("shape", [(3, 20, 30, 7), (70, 38, 63, 7), (2, 9, 14, 3)], ids=str)
def test_negative_extent_rois(shape, impl, oracle, ref_modules):
    """RoIs of negative height / width against the oracle AND the reference's kernels."""
    from detect_to_track.models import _ext
    C, H, W, k = shape
    rng = np.random.default_rng(C * 100 + k)
    rois = np.asarray(NEGATIVE_ROIS + ADVERSARIAL_ROIS[:3], np.float32)
    fm = rng.random((C, H, W), dtype=np.float32)
    gout = rng.random((len(rois), C, k, k), dtype=np.float32)
    want = oracle.roipool_fwd(fm, rois, k)
    out = _n(_ext.roipool_forward(_t(fm), _t(rois), k, impl))
    np.testing.assert_array_equal(np.isnan(out), np.isnan(want))
    np.testing.assert_allclose(np.nan_to_num(out), np.nan_to_num(want), **TOL32)
    if k == 7:
        np.testing.assert_array_equal(out, want)                 # (k = 7: every selector is bit-exact)
    want_g = oracle.roipool_bwd(gout, rois, H, W)
    assert want_g.sum() > 0                                      # these RoIs do deposit gradient
    gin = _n(_ext.roipool_backward(_t(gout), _t(rois), H, W, impl))
    np.testing.assert_allclose(gin, want_g, **TOL32)
    ref = _n(ref_modules[1].roipool_backward(_t(gout), _t(rois), H, W))
    np.testing.assert_allclose(gin, ref, **TOL32)
    np.testing.assert_array_equal(_n(_ext.roipool_bins(_t(rois), H, W, k)), oracle.roipool_bins(rois, H, W, k))


@pytest.mark.parametrize("impl", [1, 2], ids=["generic", "tuned"])
def test_nonfinite_map(impl, oracle):
    """Inf / NaN in the feature map: only bins that contain the bad pixel may be non-finite (a
    summed-area table alone would poison everything below-right of it)."""
    from detect_to_track.models import _ext
    R, C, H, W, k = 48, 6, 38, 63, 7
    rng = np.random.default_rng(3)
    fm = rng.random((C, H, W), dtype=np.float32)
    fm[1, 4, 5] = np.inf
    fm[2, 20, 30] = np.nan
    fm[3, 0, 0] = -np.inf
    fm[4, 10, 10] = np.inf
    fm[4, 30, 50] = -np.inf                                      # +inf and -inf in one channel
    rois = random_rois(R, 11)
    want = oracle.roipool_fwd(fm, rois, k)
    out = _n(_ext.roipool_forward(_t(fm), _t(rois), k, impl))
    np.testing.assert_array_equal(np.isnan(out), np.isnan(want))
    np.testing.assert_array_equal(np.isposinf(out), np.isposinf(want))
    np.testing.assert_array_equal(np.isneginf(out), np.isneginf(want))
    fin = np.isfinite(want)
    np.testing.assert_array_equal(out[fin], want[fin])           # k = 7: bit-exact, tuned or generic


@pytest.mark.parametrize("case", [(300, 1024, 38, 63), (8, 1891, 38, 75), (33, 13, 38, 63), (1, 8, 5, 9), (150, 9, 255, 18), (64, 1030, 20, 31)], ids=str)
def test_tuned_forward_is_the_generic_kernels_bits(case):
    """Round 6: the tuned k = 7 forward (d2t_roipool_fwd_direct.hip: planes of 8 channels interleaved in LDS, thread = (RoI, bin), row-major running
    sums, IEEE divide) against the thread-per-output anchor -- bit for bit, NaN pattern included, on random + adversarial + negative RoIs, signed
    data, a partial channel group, a partial pixel group and the tallest map the kernel takes."""
    from detect_to_track.models import _ext
    R, C, H, W = case
    rng = np.random.default_rng(R * 31 + C)
    extra = np.asarray(ADVERSARIAL_ROIS + NEGATIVE_ROIS, np.float32)
    rois = random_rois(R, R + C)
    rois[: min(R, len(extra))] = extra[: min(R, len(extra))]
    fm = rng.standard_normal((C, H, W)).astype(np.float32) * np.float32(100.0)
    want = _ext.roipool_forward(_t(fm), _t(rois), 7, 1)
    for impl in (0, 2):
        got = _ext.roipool_forward(_t(fm), _t(rois), 7, impl)
        assert torch.equal(got.isnan(), want.isnan())
        assert torch.equal(torch.nan_to_num(got).view(torch.int32), torch.nan_to_num(want).view(torch.int32)), (case, impl)


@pytest.mark.parametrize("case", [(64, 70, 38, 63), (300, 40, 20, 100)], ids=str)
def test_nonfinite_gradout(case):
    """Inf / NaN in gradOut.  The backward GEMM (d2t_pool_bwd.hip) multiplies every slot by a 0/1 weight for all 16 columns of a
    tile: 0 x Inf = NaN would reach columns outside the value's bin, which the reference (roipool_cuda.cu:111-117) never
    touches; a task that stored a non-finite value is recomputed with exact membership.  Pattern and finite values as the
    type-generic kernel (the reference's form)."""
    from detect_to_track.models import _ext
    R, C, H, W = case
    rng = np.random.default_rng(R)
    rois = _t(random_rois(R, 5))
    gout = rng.standard_normal((R, C, 7, 7)).astype(np.float32)
    gout[3, 5, 2, 2] = np.inf; gout[R - 1, C - 1, 6, 0] = -np.inf; gout[R // 2, 17, 0, 6] = np.nan; gout[7, 0, 3, 3] = 3e38; gout[8, 0, 3, 3] = 3e38
    got = _ext.roipool_backward(_t(gout), rois, H, W, 2)
    want = _ext.roipool_backward(_t(gout), rois, H, W, 1)
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    assert torch.equal(torch.isposinf(got), torch.isposinf(want)) and torch.equal(torch.isneginf(got), torch.isneginf(want))
    fin = torch.isfinite(want)
    torch.testing.assert_close(got[fin], want[fin], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", [(300, 130, 38, 63, 6, "random"), (40, 1030, 20, 31, 3, "random"), (17, 5, 9, 14, 14, "random"), (9, 3, 7, 5, 1, "random"),
                                  (300, 64, 38, 63, 8, "adversarial"), (30, 7, 20, 30, 5, "negative"), (520, 9, 12, 17, 2, "random"),
                                  (12, 6, 20, 20, 3, "huge")], ids=str)
def test_outside_the_envelope_backward_equals_generic_kernel(case, dtype):
    """Bin counts other than 7 (and all of f64) take the per-pixel bin lists of d2t_pool_lists.hip under the default dispatch: the same
    terms gradOut / n in the same ascending (r, i, j) order as the thread-per-pixel kernel (roipool_cuda.cu:112-124, gather form), so
    the gradient is bit-identical to D2T_IMPL_GENERIC, which test_matches_oracle pins to the oracle.  "huge": RoIs many times the
    map, whose lists exceed the workspace -- the device-side fallback to the thread-per-pixel kernel.  Non-finite gradOut included."""
    from detect_to_track.models import _ext
    R, C, H, W, k, kind = case
    rng = np.random.default_rng(R * 31 + k)
    if kind == "random":
        rois = random_rois(R, k)
    elif kind == "adversarial":
        rois = np.resize(ADVERSARIAL_ROIS, (R, 4))
    elif kind == "negative":
        rois = np.resize(NEGATIVE_ROIS, (R, 4))
    else:
        rois = np.concatenate([rng.random((R, 2)), 20 + 30 * rng.random((R, 2))], axis=1)
    gout = rng.standard_normal((R, C, k, k))
    gout[0, 0, 0, 0] = np.inf
    gout[R - 1, C - 1, k - 1, k - 1] = np.nan
    g, r = _t(gout.astype(dtype)), _t(np.asarray(rois).astype(dtype))
    got = _ext.roipool_backward(g, r, H, W, 0)
    want = _ext.roipool_backward(g, r, H, W, 1)
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    fin = ~torch.isnan(want)
    assert torch.equal(got[fin], want[fin])


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 8, 9, 11, 14, 16])
def test_forward_any_bin_count_summed_area_tables(k):
    """Bin counts 1..16 other than 7 run the interleaved summed-area kernel with k at run time (d2t_pool_tuned.hip,
    k_roipool_fwd_sat2<0>): NaN pattern of the generic kernel (empty bins: 0/0, roipool_cuda.cu:61) bit for bit, values within 1e-5;
    adversarial + random RoIs, an odd channel count, and a map with Inf / NaN in it (a poisoned table: the affected bins are redone in the
    reference's form, so they match the generic kernel's non-finite pattern exactly)."""
    from detect_to_track.models import _ext
    R, C, H, W = 90, 5, 38, 63
    rng = np.random.default_rng(k)
    rois = np.concatenate([np.asarray(ADVERSARIAL_ROIS, np.float32), np.asarray(NEGATIVE_ROIS, np.float32),
                           random_rois(R - len(ADVERSARIAL_ROIS) - len(NEGATIVE_ROIS), k)])
    fm = rng.standard_normal((C, H, W)).astype(np.float32)
    for poison in (False, True):
        if poison:
            fm[1, 7, 9] = np.inf; fm[3, 30, 50] = np.nan; fm[4, 0, 0] = -np.inf
        got = _n(_ext.roipool_forward(_t(fm), _t(rois), k, 0))
        want = _n(_ext.roipool_forward(_t(fm), _t(rois), k, 1))
        assert got.shape == (R, C, k, k)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_array_equal(np.isinf(got), np.isinf(want))
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-5, atol=1e-5)


def test_config3_properties():
    """R=300 C=1024 38x63 k=7 (BASELINE.json config 3): size-independent properties."""
    from detect_to_track.models import _ext
    R, C, H, W, k = 300, 1024, 38, 63, 7
    rois = _t(random_rois(R, 0))
    # pooling a constant map gives the constant wherever the bin is non-empty
    out = _ext.roipool_forward(torch.full((C, H, W), 3.0, device=DEV), rois, k)
    bins = _ext.roipool_bins(rois, H, W, k)
    empty = ((bins[..., 1] <= bins[..., 0]) | (bins[..., 3] <= bins[..., 2]))[:, None].expand_as(out)
    assert torch.equal(out.isnan(), empty)              # 0/0 exactly on the empty bins
    assert torch.allclose(out[~empty], torch.full_like(out[~empty], 3.0), rtol=1e-6)
    # adjointness <out, G> == <FM, gFM>; mass conservation sum(gFM) == sum(G) for non-empty bins
    torch.manual_seed(5)
    fm = torch.rand(C, H, W, device=DEV)
    G = torch.rand(R, C, k, k, device=DEV)
    out = _ext.roipool_forward(fm, rois, k)
    gin = _ext.roipool_backward(G, rois, H, W)
    Gv = torch.where(empty, torch.zeros_like(G), G).double()       # empty bins take no gradient
    assert torch.allclose((torch.nan_to_num(out).double() * Gv).sum(), (fm.double() * gin.double()).sum(), rtol=1e-5)
    assert torch.allclose(gin.double().sum(), Gv.sum(), rtol=1e-5)
    # atomic-free backward: identical bits on a second run
    assert torch.equal(gin, _ext.roipool_backward(G, rois, H, W))
    # channel independence: a 64-channel slice pools to the same bits
    sl = _ext.roipool_forward(fm[128:192].contiguous(), rois, k)
    ref_sl = out[:, 128:192]
    assert bool(((sl == ref_sl) | (sl.isnan() & ref_sl.isnan())).all())


def test_empty_and_errors():
    from detect_to_track.models import ROIPool
    rp = ROIPool(7)
    out = rp(torch.rand(4, 10, 10, device=DEV), torch.empty(0, 4, device=DEV))
    assert out.shape == (0, 4, 7, 7)
    fm = torch.rand(4, 10, 10, device=DEV).requires_grad_(True)
    rp(fm, torch.tensor([[0.5, 0.5, 0.4, 0.4]], device=DEV)).sum().backward()
    assert fm.grad.shape == fm.shape
    with pytest.raises(RuntimeError, match="CPU op not implemented"):
        rp(torch.rand(4, 10, 10), torch.rand(1, 4))
    with pytest.raises(RuntimeError, match="CPU op not implemented"):
        rp(torch.rand(4, 10, 10, device=DEV), torch.rand(1, 4))
    with pytest.raises(RuntimeError):
        rp(torch.rand(4, 10, 10, device=DEV), torch.rand(1, 4, device=DEV).double())
