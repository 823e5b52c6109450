"""Region proposal network head (Faster R-CNN, arXiv 1506.01497); interface of reference models/rpn.py:9-52."""
from typing import Tuple

from torch import Tensor, nn
from torch.nn.functional import relu, softmax


class RPN(nn.Module):
    """3x3 conv to 512 channels, then 1x1 objectness (2 per anchor) and box-offset (4 per anchor) maps.

    Args:
        in_channels: input feature map channels.
        n_anchors: anchors per feature map cell.
    """

    def __init__(self, in_channels: int, n_anchors: int) -> None:
        super().__init__()
        self.conv = nn.Conv2d(in_channels, 512, kernel_size=3, padding=1)
        self.cls_fc = nn.Conv2d(512, 2 * n_anchors, kernel_size=1)
        self.reg_fc = nn.Conv2d(512, 4 * n_anchors, kernel_size=1)

    @staticmethod
    def _per_anchor(x: Tensor, width: int) -> Tensor:
        """(B, a*width, H, W) -> (B, H*W*a, width): the values of one anchor stay together."""
        return x.permute(0, 2, 3, 1).reshape(x.size(0), -1, width)

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """x: (B, C, H, W) -> (o_hat (B, |A|, 2) softmaxed object / not object, b_hat (B, |A|, 4),
        the (B, 512, H, W) regression features the tracker also reads)."""
        x = relu(self.conv(x))
        o_hat = softmax(self._per_anchor(self.cls_fc(x), 2), dim=2)
        b_hat = self._per_anchor(self.reg_fc(x), 4)
        return o_hat, b_hat, x
