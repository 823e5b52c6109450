"""ROIPool: AVERAGE pooling of a (C, H, W) map over k x k bins of each RoI.

Interface of reference roipool/roipool.py:22-81; arithmetic in libd2t_ops.so via ``_ext``.

RoIs are (centre_i, centre_j, height, width) as fractions of the map.  Reference behaviour kept:
only the RoI's top-left corner is clamped to the map, so a box overhanging the top/left edge is
shifted rather than cropped; an empty bin averages 0/0 = NaN (roipool_cuda.cu:41-42,61).
"""
from typing import Optional, Tuple

from torch import Tensor
from torch.autograd import Function
from torch.nn import Module

from .. import _ext


class ROIPoolFunction(Function):
    @staticmethod
    def forward(ctx, FM: Tensor, rois: Tensor, r_hw: int) -> Tensor:
        # (C, H, W), (|R|, 4) -> (|R|, C, r_hw, r_hw)
        out = _ext.roipool_forward(FM, rois, r_hw)
        ctx.save_for_backward(rois)
        ctx.i_h, ctx.i_w = FM.shape[-2:]
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
        (rois,) = ctx.saved_tensors
        grad_fm = _ext.roipool_backward(grad_out.contiguous(), rois, ctx.i_h, ctx.i_w)
        return grad_fm, None, None          # no gradient flows to the boxes


class ROIPool(Module):
    """Average RoI pooling (arXiv 1504.08083 geometry, mean instead of max).

    Args:
        r_hw: pooled height and width; r_hw**2 bins per RoI.
    """

    def __init__(self, r_hw: int) -> None:
        super().__init__()
        self.r_hw = r_hw

    def forward(self, FM: Tensor, rois: Tensor) -> Tensor:
        """FM: (C, H, W); rois: (|R|, 4) ijhw fractional -> (|R|, C, r_hw, r_hw)."""
        return ROIPoolFunction.apply(FM, rois, self.r_hw)

    def extra_repr(self) -> str:
        return f"r_hw={self.r_hw}"
