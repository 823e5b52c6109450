"""R-FCN detection head (arXiv 1605.06409) on top of PSROIPool; interface of reference models/rfcn.py:10-84."""
from typing import Tuple

from torch import Tensor, nn

from .ps_roipool.ps_roipool import PSROIPool


class _RFCNHead(nn.Module):
    """1x1 conv to n_targets*k^2 position-sensitive score maps, PSROIPool, vote (mean over the k x k grid).

    Args:
        in_channels: input feature map channels.
        n_targets: classes + 1 for the classifier, 4 for the box regressor.
        k: spatial grid height and width.
    """

    def __init__(self, in_channels: int, n_targets: int, k: int) -> None:
        super().__init__()
        self.sm_conv = nn.Conv2d(in_channels, n_targets * k ** 2, kernel_size=1)
        self.roi_pool = PSROIPool(n_targets, k)
        self.n_targets = n_targets

    def forward(self, x: Tensor, regions: Tensor) -> Tensor:
        """x: (C, H, W); regions: (|R|, 4) ijhw fractions -> (|R|, n_targets) scores."""
        score_map = self.sm_conv(x[None]).squeeze(0)                 # (n_targets*k^2, H, W)
        pooled = self.roi_pool(score_map, regions)                   # (|R|, n_targets, k, k)
        return pooled.mean(-1).mean(-1)


class RFCN(nn.Module):
    """Dilated 3x3 channel reduction to 512, then a classification and a box-regression head.

    Args:
        in_channels: input feature map channels.
        n_classes: number of non-background classes.
        k: spatial grid height and width.
    """

    def __init__(self, in_channels: int, n_classes: int, k: int) -> None:
        super().__init__()
        self.channel_reduce = nn.Conv2d(in_channels, 512, kernel_size=3, dilation=6, padding=6)
        self.cls_head = _RFCNHead(512, n_classes + 1, k)
        self.reg_head = _RFCNHead(512, 4, k)
        self.relu = nn.ReLU(inplace=True)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, x: Tensor, regions: Tensor) -> Tuple[Tensor, Tensor]:
        """x: (C, H, W); regions: (|R|, 4) -> (c_hat (|R|, n_classes + 1) softmaxed, b_hat (|R|, 4))."""
        x = self.relu(self.channel_reduce(x[None])).squeeze(0)       # (512, H, W)
        return self.softmax(self.cls_head(x, regions)), self.reg_head(x, regions)
