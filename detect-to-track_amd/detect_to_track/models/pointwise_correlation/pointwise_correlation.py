"""PointwiseCorrelation: sliding-window cross-frame feature correlation (D&T, arXiv 1710.03958).

Interface of reference pointwise_correlation/pointwise_correlation.py:25-95; the arithmetic
runs in libd2t_ops.so (gfx950 HIP) through ``_ext``.

Semantics worth knowing (they are the reference's, reproduced on purpose):
for a centre pixel (i, j) the displaced rows run ``max(0, i-d) .. min(i+d, H)-1`` in steps of
``stride`` -- the upper bound is exclusive, so displacement ``+d`` is never produced and the last
row / column of every (2d+1, 2d+1) map is zero; near the top/left border the stride phase is
anchored at pixel 0 (reference pointwise_correlation_cuda.cu:92-93).
"""
from typing import Optional, Tuple

from torch import Tensor
from torch.autograd import Function
from torch.nn import Module

from .. import _ext


class PointwiseCorrelationFunction(Function):
    """out[b, i, j, di-i+d, dj-j+d] = <FM0[b, :, i, j], FM1[b, :, di, dj]>."""

    @staticmethod
    def forward(ctx, FM0: Tensor, FM1: Tensor, d_max: int, stride: int) -> Tensor:
        # (B, C, H, W) x 2  ->  (B, H, W, 2d+1, 2d+1)
        out = _ext.pointwise_correlation_forward(FM0, FM1, d_max, stride)
        ctx.save_for_backward(FM0, FM1)
        ctx.d_max, ctx.stride = d_max, stride
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor) -> Tuple[Tensor, Tensor, Optional[Tensor], Optional[Tensor]]:
        # both feature-map gradients are always produced, as in the reference (:63-67)
        FM0, FM1 = ctx.saved_tensors
        g0, g1 = _ext.pointwise_correlation_backward(grad_out.contiguous(), FM0, FM1, ctx.d_max, ctx.stride)
        return g0, g1, None, None


class PointwiseCorrelation(Module):
    """Local correlation of two feature maps over displacements in [-d_max, d_max).

    Args:
        d_max: maximum displacement; the output map is (2*d_max+1) square.
        stride: step between sampled displacements.
    """

    def __init__(self, d_max: int, stride: int) -> None:
        super().__init__()
        self.d_max = d_max
        self.stride = stride

    def forward(self, FM0: Tensor, FM1: Tensor) -> Tensor:
        """FM0, FM1: (B, C, H, W) at times t and t+tau -> (B, H, W, 2d+1, 2d+1)."""
        return PointwiseCorrelationFunction.apply(FM0, FM1, self.d_max, self.stride)

    def extra_repr(self) -> str:
        return f"d_max={self.d_max}, stride={self.stride}"
