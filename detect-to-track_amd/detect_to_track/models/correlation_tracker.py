"""CorrelationTracker: the caller of PointwiseCorrelation (x3) and ROIPool in the reference's model
graph (reference models/correlation_tracker.py:13-87), with the glue between the two ops fused.

The reference computes three correlations with three calls, turns each (1, H, W, 2d+1, 2d+1)
result into ((2d+1)^2, H, W) with view + permute (three copies), and ``torch.cat``'s them behind the
two RPN feature maps (a fourth, 21.6 MB copy) before ROIPool (:64-82).  Here ``TrackFeaturesFunction``
allocates the (2*Cr + 3*(2d+1)^2, H, W) buffer once, copies the two RPN maps into its head and lets
ONE call of the HIP library write all three correlations channel-major straight into its tail
(d2t_corr_fwd_levels_f32); the backward reads the buffer's gradient in place.  Values are
bit-identical to the unfused composition.  Same constructor, attributes and forward signature as
the reference module.

``fast_forward`` (constructor keyword / attribute, default False) opts in to D2T_IMPL_FAST: the 1024- and
2048-channel levels of a B = 1 pair then split their channels over workgroups (fused forward 170 -> 128 us);
deterministic and within 1e-5 of the reference's values, not bit-identical to them.
"""
from typing import Mapping, Optional, Tuple

import torch
from torch import Tensor, nn
from torch.autograd import Function

from . import _ext, _native
from .roipool.roipool import ROIPool


class TrackFeaturesFunction(Function):
    """(reg_fm_0, reg_fm_1, [FM0_l, FM1_l] x L) -> (2*Cr + L*(2d+1)^2, H, W), the tensor the reference
    builds at correlation_tracker.py:64-80.  FM*_l: (1, C_l, H, W) float32.
    Batched over the P pairs of a step: reg_fm_* (P, Cr, H, W), FM*_l (P, C_l, H, W) -> (P, 2*Cr + L*(2d+1)^2, H, W) with
    ONE call of the library for all pairs and levels (the reference loops over pairs, trainer.py:263-264)."""

    @staticmethod
    def forward(ctx, d_max: int, stride: int, impl: int, reg_fm_0: Tensor, reg_fm_1: Tensor, *fms: Tensor) -> Tensor:
        if len(fms) % 2 or not fms:
            raise RuntimeError("feature maps come in (FM0, FM1) pairs")
        fm0s, fm1s = [f.contiguous() for f in fms[0::2]], [f.contiguous() for f in fms[1::2]]
        batched = reg_fm_0.dim() == 4
        if not batched:
            reg_fm_0, reg_fm_1 = reg_fm_0[None], reg_fm_1[None]
        cr = reg_fm_0.size(1)
        cells = (2 * d_max + 1) ** 2
        P, _, H, W = fm0s[0].shape
        buf = torch.empty((P, 2 * cr + len(fm0s) * cells, H, W), dtype=reg_fm_0.dtype, device=reg_fm_0.device)
        buf[:, :cr] = reg_fm_0
        buf[:, cr:2 * cr] = reg_fm_1
        _ext.pointwise_correlation_levels_forward(fm0s, fm1s, d_max, stride, out=(buf, 2 * cr), impl=impl)
        ctx.save_for_backward(*fm0s, *fm1s)
        ctx.meta = (d_max, stride, cr, len(fm0s), batched)
        return buf if batched else buf[0]

    @staticmethod
    def backward(ctx, grad: Tensor) -> Tuple[Optional[Tensor], ...]:
        d_max, stride, cr, L, batched = ctx.meta
        fm0s, fm1s = list(ctx.saved_tensors[:L]), list(ctx.saved_tensors[L:])
        grad = grad.contiguous()
        gb = grad if batched else grad[None]
        g0, g1 = _ext.pointwise_correlation_levels_backward(gb, 2 * cr, fm0s, fm1s, d_max, stride)
        grads = [g for pair in zip(g0, g1) for g in pair]
        return (None, None, None, gb[:, :cr] if batched else grad[:cr], gb[:, cr:2 * cr] if batched else grad[cr:2 * cr], *grads)


class CorrelationTracker(nn.Module):
    """Given features from time steps t and t+tau, predict box transformations between them
    (D&T, arXiv 1710.03958).  Interface of reference correlation_tracker.py:13-87.

    Args:
        d_max: maximum displacement for pointwise correlations.
        r_hw: height and width of pooled feature maps.
        reg_channels: RPN feature map channels.
        stride: correlation stride.
        fast_forward: (not in the reference) let the fused forward split channels over workgroups (D2T_IMPL_FAST).
    """

    def __init__(self, d_max: int, r_hw: int, reg_channels: int, stride: int = 1, fast_forward: bool = False) -> None:
        super().__init__()
        self.d_max, self.stride, self.fast_forward = d_max, stride, bool(fast_forward)
        self.pool = ROIPool(r_hw)
        self.fc_channels = (3 * pow(2 * d_max + 1, 2) + 2 * reg_channels) * pow(r_hw, 2)
        self.reg_fc = nn.Linear(self.fc_channels, 4)

    def track_features(self, fm_pyr_0: Mapping[str, Tensor], fm_pyr_1: Mapping[str, Tensor],
                       reg_fm_0: Tensor, reg_fm_1: Tensor) -> Tensor:
        keys = ["c3", "c4", "c5"]
        c3_0, c4_0, c5_0 = [fm_pyr_0[k][None] for k in keys]
        c3_1, c4_1, c5_1 = [fm_pyr_1[k][None] for k in keys]
        # c3 has half the stride of c4 and c5 (reference :60-61)
        c3_0 = nn.functional.interpolate(c3_0, scale_factor=1 / 2)
        c3_1 = nn.functional.interpolate(c3_1, scale_factor=1 / 2)
        impl = _native.IMPL_FAST if self.fast_forward else _native.IMPL_AUTO
        return TrackFeaturesFunction.apply(self.d_max, self.stride, impl, reg_fm_0, reg_fm_1,
                                           c3_0, c3_1, c4_0, c4_1, c5_0, c5_1)

    def forward_pairs(self, fm_pyrs_0: Mapping[str, Tensor], fm_pyrs_1: Mapping[str, Tensor],
                      reg_fm_0: Tensor, reg_fm_1: Tensor, rois: "list[Tensor]") -> "list[Tensor]":
        """The P pairs of a step at once (not in the reference, whose trainer loops over pairs: trainer.py:263-264).
        fm_pyrs_*: {"c3","c4","c5"} of (P, C, H, W) maps at t / t+tau; reg_fm_*: (P, Cr, H, W); rois: P tensors (|R_p|, 4).
        ONE fused correlation call for all pairs and levels (B = P fills the chip where B = 1 leaves most of it idle), then
        ROIPool and the FC per pair on slices of the one buffer.  Pair p's result equals forward() on pair p."""
        c3_0 = nn.functional.interpolate(fm_pyrs_0["c3"], scale_factor=1 / 2)
        c3_1 = nn.functional.interpolate(fm_pyrs_1["c3"], scale_factor=1 / 2)
        impl = _native.IMPL_FAST if self.fast_forward else _native.IMPL_AUTO
        feats = TrackFeaturesFunction.apply(self.d_max, self.stride, impl, reg_fm_0.contiguous(), reg_fm_1.contiguous(),
                                            c3_0, c3_1, fm_pyrs_0["c4"], fm_pyrs_1["c4"], fm_pyrs_0["c5"], fm_pyrs_1["c5"])
        out = []
        for p, r in enumerate(rois):
            pooled = self.pool(feats[p], r)
            out.append(self.reg_fc(pooled.view(pooled.size(0), self.fc_channels)))
        return out

    def forward(self, fm_pyr_0: Mapping[str, Tensor], fm_pyr_1: Mapping[str, Tensor],
                reg_fm_0: Tensor, reg_fm_1: Tensor, rois: Tensor) -> Tensor:
        """fm_pyr_*: {"c3","c4","c5"} backbone maps at t / t+tau; reg_fm_*: (Cr, H, W) RPN features;
        rois: (|R|, 4) -> t_hat (|R|, 4)."""
        track_feats = self.track_features(fm_pyr_0, fm_pyr_1, reg_fm_0, reg_fm_1)   # (3*(2d+1)^2 + 2Cr, H, W)
        pooled = self.pool(track_feats, rois)                                        # (|R|, C, rHW, rHW)
        return self.reg_fc(pooled.view(pooled.size(0), self.fc_channels))
