"""DetectTrackModule: the container of the four sub-networks (reference models/detect_track.py:11-61).

Like the reference's it has no forward of its own: the training step (reference trainer.py:133-256)
and the detector drive the parts.  ``bench_model.py`` at the repository root runs that training step
on synthetic frames (BASELINE.json config 4)."""
from torch.nn import Module

from .correlation_tracker import CorrelationTracker
from .resnet import resnet_backbone
from .rfcn import RFCN
from .rpn import RPN


class DetectTrackModule(Module):
    """backbone: images -> {"c3","c4","c5"}; rpn: c4 -> proposals + regression features; rcnn: c5,
    regions -> classes and offsets; c_tracker: both frames' maps -> box transformations."""

    stage3_outchannels = 512
    stage4_outchannels = 1024
    stage5_outchannels = 2048

    def __init__(self, backbone_arch: str, first_trainable_stage: int, n_anchors: int, n_classes: int,
                 k: int, d_max: int, r_hw: int) -> None:
        super().__init__()
        self.backbone = resnet_backbone(backbone_arch, first_trainable_stage)
        self.rpn = RPN(self.stage4_outchannels, n_anchors)
        self.rcnn = RFCN(self.stage5_outchannels, n_classes, k)
        self.c_tracker = CorrelationTracker(d_max, r_hw, self.rpn.conv.out_channels)

    def forward(self):
        raise NotImplementedError("forward passes are driven by the trainer / detector, as in the reference")
