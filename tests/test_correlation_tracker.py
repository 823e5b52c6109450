"""GPU tests of the fused tracker glue: several correlations in one call, written
channel-major straight into the concatenation buffer ROIPool reads.

Parity = bit-equality with the composition the reference spells out (correlation_tracker.py:64-83):
three PointwiseCorrelation calls, view + permute, torch.cat -- built here from the (already pinned)
single-call ops of this package.  The C ABI entry points d2t_corr_{fwd,bwd}_levels_f32 are reached
through _ext.pointwise_correlation_levels_{forward,backward}.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


MFMA = 2                                                                     # D2T_IMPL_MFMA: tuned kernels, never channel-split
FAST = 5                                                                     # D2T_IMPL_FAST: the forward may split channels over workgroups


def _splits(Cs, B, H, W, d, s):
    import ctypes
    from detect_to_track.models import _native
    return _native.lib.d2t_corr_fwd_levels_workspace_bytes(len(Cs), (ctypes.c_int * len(Cs))(*Cs), B, H, W, d, s) > 0


def _unfused(fm0s, fm1s, d, s, impl=0):
    from detect_to_track.models import _ext
    outs = []
    for a, b in zip(fm0s, fm1s):
        cf = _ext.pointwise_correlation_forward(a, b, d, s, impl)            # (B,H,W,cw,cw)
        outs.append(cf.view(cf.size(0), cf.size(1), cf.size(2), -1).permute(0, 3, 1, 2))
    return torch.cat(outs, 1)                                                # (B, L*cells, H, W)


CASES = [  # (B, Cs, H, W, d, s)
    (1, (32, 48, 80), 38, 75, 8, 1),        # the tracker's shape class: one launch of 1-tile workgroups
    (1, (512, 1024, 2048), 38, 75, 8, 1),   # the model's real channel counts (correlation_tracker.py:68-70)
    (2, (20, 7), 19, 33, 8, 1),             # B > 1: batch stride wider than one level
    (3, (48,), 38, 63, 8, 1),               # 2-tile segments
    (8, (40, 24), 38, 63, 8, 1),            # full-grid kernels, one launch per level
    (2, (6, 9, 4, 5), 11, 13, 3, 2),        # generic kernels (d != 8): four levels
    (1, (5,), 9, 21, 8, 1),
]


@pytest.mark.parametrize("case", CASES, ids=str)
def test_levels_forward_backward_equal_unfused(case):
    from detect_to_track.models import _ext
    B, Cs, H, W, d, s = case
    cells = (2 * d + 1) ** 2
    g = torch.Generator().manual_seed(sum(Cs) * 131 + H)
    fm0s = [torch.rand(B, C, H, W, generator=g).to(DEV) for C in Cs]
    fm1s = [torch.rand(B, C, H, W, generator=g).to(DEV) for C in Cs]
    want = _unfused(fm0s, fm1s, d, s, MFMA if d == 8 and s == 1 and W >= 20 else 0)   # unsplit kernels: the reference's chain, bit for bit
    got = _ext.pointwise_correlation_levels_forward(fm0s, fm1s, d, s)
    assert got.shape == want.shape
    assert torch.equal(got, want)                                             # default dispatch: the reference's chain, bit for bit
    if _splits(Cs, B, H, W, d, s):
        # opt-in (D2T_IMPL_FAST): channels split over workgroups, partial sums added in a fixed order -- same terms, other association
        fast = _ext.pointwise_correlation_levels_forward(fm0s, fm1s, d, s, impl=FAST)
        torch.testing.assert_close(fast, want, rtol=1e-5, atol=1e-5)
        assert not torch.equal(fast, want) or max(Cs) < 640                   # it really took the other path
        assert torch.equal(fast, _ext.pointwise_correlation_levels_forward(fm0s, fm1s, d, s, impl=FAST))   # deterministic
    else:
        assert torch.equal(_ext.pointwise_correlation_levels_forward(fm0s, fm1s, d, s, impl=FAST), want)
    # into the middle of a wider buffer (a torch.cat target), neighbours untouched
    pad0, pad1 = 5, 3
    buf = torch.full((B, pad0 + len(Cs) * cells + pad1, H, W), -7.0, device=DEV)
    _ext.pointwise_correlation_levels_forward(fm0s, fm1s, d, s, out=(buf, pad0))
    assert torch.equal(buf[:, pad0:pad0 + len(Cs) * cells], want)
    assert bool((buf[:, :pad0] == -7).all()) and bool((buf[:, pad0 + len(Cs) * cells:] == -7).all())
    # backward from the buffer's gradient, in place
    gbuf = torch.rand(B, pad0 + len(Cs) * cells + pad1, H, W, generator=g).to(DEV)
    g0, g1 = _ext.pointwise_correlation_levels_backward(gbuf, pad0, fm0s, fm1s, d, s)
    for l, (a, b) in enumerate(zip(fm0s, fm1s)):
        gl = gbuf[:, pad0 + l * cells: pad0 + (l + 1) * cells].permute(0, 2, 3, 1).reshape(B, H, W, 2 * d + 1, 2 * d + 1)
        r0, r1 = _ext.pointwise_correlation_backward(gl.contiguous(), a, b, d, s)
        # Same kernels either way: where the reference-layout call takes the 8-wave strip kernel (d = 8, stride 1, a map of at
        # least 17 x 20) the channel-major call first re-lays the gradient into the workspace d2t_corr_bwd_levels_workspace_bytes
        # asks for and runs that kernel on the copy; elsewhere both run the 16- / 4-wave kernels -- bit-equal in every case.
        assert torch.equal(g0[l], r0) and torch.equal(g1[l], r1)


def test_levels_match_oracle(oracle):
    """The fused entry against the CPU oracle directly (not only against our own single-call op)."""
    from detect_to_track.models import _ext
    rng = np.random.default_rng(5)
    B, Cs, H, W, d = 1, (24, 40), 38, 75, 8
    fm0 = [rng.random((B, C, H, W), dtype=np.float32) for C in Cs]
    fm1 = [rng.random((B, C, H, W), dtype=np.float32) for C in Cs]
    got = _ext.pointwise_correlation_levels_forward([torch.from_numpy(a).to(DEV) for a in fm0],
                                                    [torch.from_numpy(a).to(DEV) for a in fm1], d, 1).cpu().numpy()
    for l in range(len(Cs)):
        want = oracle.corr_fwd(fm0[l], fm1[l], d, 1).reshape(B, H, W, 289).transpose(0, 3, 1, 2)
        np.testing.assert_array_equal(got[:, l * 289:(l + 1) * 289], want)


def test_levels_backward_matches_oracle(oracle):
    """d2t_corr_bwd_levels_f32 (channel-major gradient read in place from a wider buffer) against the CPU oracle
    directly: pointwise_correlation_cuda.cu:145-171 per level, terms summed in double, 1e-5 of sum|terms|."""
    from detect_to_track.models import _ext
    rng = np.random.default_rng(11)
    for B, Cs, H, W in [(1, (24, 40, 72), 38, 75), (2, (300,), 38, 63)]:   # one shared 4-wave launch; the 256-channel strip kernels
        d, cells, pad0 = 8, 289, 3
        fm0 = [rng.random((B, C, H, W), dtype=np.float32) for C in Cs]
        fm1 = [rng.random((B, C, H, W), dtype=np.float32) for C in Cs]
        gbuf = rng.standard_normal((B, pad0 + len(Cs) * cells + 2, H, W)).astype(np.float32)
        g0, g1 = _ext.pointwise_correlation_levels_backward(torch.from_numpy(gbuf).to(DEV), pad0,
                                                            [torch.from_numpy(a).to(DEV) for a in fm0],
                                                            [torch.from_numpy(a).to(DEV) for a in fm1], d, 1)
        for l in range(len(Cs)):
            gl = np.ascontiguousarray(gbuf[:, pad0 + l * cells: pad0 + (l + 1) * cells].transpose(0, 2, 3, 1)).reshape(B, H, W, 17, 17)
            (w0, w1), (m0, m1) = oracle.corr_bwd_acc64(gl, fm0[l], fm1[l], d, 1)
            oracle.assert_within_contract(g0[l].cpu().numpy(), w0, m0, 1e-5, f"gradFM0 level {l}")
            oracle.assert_within_contract(g1[l].cpu().numpy(), w1, m1, 1e-5, f"gradFM1 level {l}")


def _reference_composition(mod, pyr0, pyr1, reg0, reg1, rois):
    """correlation_tracker.py:55-87 spelled out with this package's single-call ops."""
    from detect_to_track.models import PointwiseCorrelation
    pc = PointwiseCorrelation(mod.d_max, mod.stride)
    c3_0, c4_0, c5_0 = [pyr0[k][None] for k in ("c3", "c4", "c5")]
    c3_1, c4_1, c5_1 = [pyr1[k][None] for k in ("c3", "c4", "c5")]
    c3_0 = torch.nn.functional.interpolate(c3_0, scale_factor=1 / 2)
    c3_1 = torch.nn.functional.interpolate(c3_1, scale_factor=1 / 2)
    feats = [cf.squeeze(0).view(cf.size(1), cf.size(2), -1).permute(2, 0, 1)
             for cf in (pc(c3_0, c3_1), pc(c4_0, c4_1), pc(c5_0, c5_1))]
    track = torch.cat([reg0, reg1, *feats])
    pooled = mod.pool(track, rois)
    return mod.reg_fc(pooled.view(pooled.size(0), mod.fc_channels))


def test_module_matches_reference_composition():
    from detect_to_track.models import CorrelationTracker
    torch.manual_seed(3)
    H, W, cr = 38, 75, 16
    mod = CorrelationTracker(8, 7, cr).to(DEV)
    assert mod.fc_channels == (3 * 289 + 2 * cr) * 49 and mod.reg_fc.in_features == mod.fc_channels

    def leaf(*shape):
        return torch.rand(*shape, device=DEV).requires_grad_(True)
    ins_a = dict(p0={"c3": leaf(12, 2 * H, 2 * W), "c4": leaf(20, H, W), "c5": leaf(28, H, W)},
                 p1={"c3": leaf(12, 2 * H, 2 * W), "c4": leaf(20, H, W), "c5": leaf(28, H, W)},
                 r0=leaf(cr, H, W), r1=leaf(cr, H, W))
    ins_b = {k: ({kk: vv.detach().clone().requires_grad_(True) for kk, vv in v.items()} if isinstance(v, dict)
                 else v.detach().clone().requires_grad_(True)) for k, v in ins_a.items()}
    rois = torch.tensor([[0.5, 0.5, 0.4, 0.3], [0.2, 0.7, 0.3, 0.5], [0.8, 0.3, 0.25, 0.6]], device=DEV)
    out_a = mod(ins_a["p0"], ins_a["p1"], ins_a["r0"], ins_a["r1"], rois)
    out_b = _reference_composition(mod, ins_b["p0"], ins_b["p1"], ins_b["r0"], ins_b["r1"], rois)
    assert out_a.shape == (3, 4) and torch.equal(out_a, out_b)
    w = torch.rand_like(out_a)
    (out_a * w).sum().backward()
    ga = [ins_a["p0"][k].grad for k in ("c3", "c4", "c5")] + [ins_a["p1"][k].grad for k in ("c3", "c4", "c5")] + \
         [ins_a["r0"].grad, ins_a["r1"].grad, mod.reg_fc.weight.grad.clone()]
    mod.zero_grad()
    (out_b * w).sum().backward()
    gb = [ins_b["p0"][k].grad for k in ("c3", "c4", "c5")] + [ins_b["p1"][k].grad for k in ("c3", "c4", "c5")] + \
         [ins_b["r0"].grad, ins_b["r1"].grad, mod.reg_fc.weight.grad.clone()]
    for x, y in zip(ga, gb):
        # the fused backward reads the channel-major gradient in place (16- / 4-wave strip kernels), the composition calls
        # PointwiseCorrelation's backward on the reference layout (8-wave kernel): same terms, other summation order
        assert x is not None
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-5 * float(y.abs().max()))


def test_forward_pairs_equals_per_pair_forward():
    """CorrelationTracker.forward_pairs (the P pairs of a step through ONE fused correlation call with B = P, SURVEY 8f-1's
    'batched B > 1 call across pairs') against forward() pair by pair: forward values bit for bit (the same kernels run per
    batch item), gradients to 1e-5 of their scale (the batched backward may take another kernel for the larger grid)."""
    from detect_to_track.models import CorrelationTracker
    torch.manual_seed(7)
    P, H, W, cr = 3, 38, 63, 16
    mod = CorrelationTracker(8, 7, cr).to(DEV)

    def leaf(*shape):
        return torch.rand(*shape, device=DEV).requires_grad_(True)
    Cs = {"c3": 12, "c4": 20, "c5": 28}
    p0 = {k: leaf(P, c, (2 * H if k == "c3" else H), (2 * W if k == "c3" else W)) for k, c in Cs.items()}
    p1 = {k: leaf(P, c, (2 * H if k == "c3" else H), (2 * W if k == "c3" else W)) for k, c in Cs.items()}
    r0, r1 = leaf(P, cr, H, W), leaf(P, cr, H, W)
    rois = [torch.tensor([[0.5, 0.5, 0.4, 0.3], [0.2, 0.7, 0.3, 0.5]], device=DEV)[: 1 + p % 2] for p in range(P)]
    outs = mod.forward_pairs(p0, p1, r0, r1, rois)
    ws = [torch.rand_like(o) for o in outs]
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    leaves = list(p0.values()) + list(p1.values()) + [r0, r1]
    g_batched = [x.grad.clone() for x in leaves] + [mod.reg_fc.weight.grad.clone()]
    for x in leaves:
        x.grad = None
    mod.zero_grad()
    singles = [mod({k: v[p] for k, v in p0.items()}, {k: v[p] for k, v in p1.items()}, r0[p], r1[p], rois[p]) for p in range(P)]
    for a, b in zip(outs, singles):
        assert a.shape == b.shape and torch.equal(a, b)
    sum((o * w).sum() for o, w in zip(singles, ws)).backward()
    g_single = [x.grad for x in leaves] + [mod.reg_fc.weight.grad]
    for x, y in zip(g_batched, g_single):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-5 * float(y.abs().max()))


def test_levels_argument_errors():
    from detect_to_track.models import _ext
    a = torch.rand(1, 4, 9, 21, device=DEV)
    with pytest.raises(RuntimeError):
        _ext.pointwise_correlation_levels_forward([a], [a, a], 8, 1)
    with pytest.raises(RuntimeError, match="CPU op not implemented"):
        _ext.pointwise_correlation_levels_forward([a.cpu()], [a.cpu()], 8, 1)
    with pytest.raises(RuntimeError):
        _ext.pointwise_correlation_levels_forward([a], [torch.rand(1, 4, 9, 20, device=DEV)], 8, 1)
    with pytest.raises(RuntimeError):                            # buffer too narrow
        _ext.pointwise_correlation_levels_forward([a], [a], 8, 1, out=(torch.empty(1, 100, 9, 21, device=DEV), 0))
    with pytest.raises(RuntimeError):                            # more than 4 levels
        _ext.pointwise_correlation_levels_forward([a] * 5, [a] * 5, 8, 1)
