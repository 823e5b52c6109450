"""PSROIPool: position-sensitive AVERAGE RoI pooling (R-FCN, arXiv 1605.06409).

Interface of reference ps_roipool/ps_roipool.py:24-99; arithmetic in libd2t_ops.so via ``_ext``.

Reference behaviour kept: output (r, t, i, j) pools channel ``(t+1) * (i*k + j)`` (NOT
``t*k*k + i*k + j``; ps_roipool_cuda.cu:58), cells are cropped to the map edge by edge, and an
empty cell yields 0 (guarded divide, :67-69).
"""
from typing import Optional, Tuple

from torch import Tensor
from torch.autograd import Function
from torch.nn import Module

from .. import _ext


class PSROIPoolFunction(Function):
    @staticmethod
    def forward(ctx, FM: Tensor, rois: Tensor, n_targets: int, r_hw: int) -> Tensor:
        # (n_targets*r_hw^2, H, W), (|R|, 4) -> (|R|, n_targets, r_hw, r_hw)
        need = n_targets * r_hw ** 2
        if FM.size(0) != need:
            raise ValueError(
                f"expected {need} feature map channels (n_targets * r_hw^2), "
                f"got a feature map of shape {tuple(FM.shape)}"
            )
        pooled = _ext.ps_roipool_forward(FM, rois, n_targets, r_hw)
        ctx.save_for_backward(rois)
        _, ctx.fm_h, ctx.fm_w = FM.shape
        return pooled

    @staticmethod
    def backward(ctx, grad_out: Tensor) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor], Optional[Tensor]]:
        (rois,) = ctx.saved_tensors
        grad_fm = _ext.ps_roipool_backward(grad_out.contiguous(), rois, ctx.fm_h, ctx.fm_w)
        return grad_fm, None, None, None


class PSROIPool(Module):
    """Position-sensitive average RoI pooling.

    Args:
        n_targets: prediction targets per RoI (classes+1 for the cls head, 4 for the box head).
        r_hw: pooled height and width.
    """

    def __init__(self, n_targets: int, r_hw: int) -> None:
        super().__init__()
        self.n_targets = n_targets
        self.r_hw = r_hw

    def forward(self, FM: Tensor, rois: Tensor) -> Tensor:
        """FM: (n_targets*r_hw^2, H, W); rois: (|R|, 4) -> (|R|, n_targets, r_hw, r_hw)."""
        return PSROIPoolFunction.apply(FM, rois, self.n_targets, self.r_hw)

    def extra_repr(self) -> str:
        return f"n_targets={self.n_targets}, r_hw={self.r_hw}"
