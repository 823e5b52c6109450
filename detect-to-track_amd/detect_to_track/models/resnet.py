"""Stride-reduced ResNet with a dilated last stage: the backbone of the reference's model graph
(reference models/resnet.py:12-39), self-contained.

The reference takes torchvision's ``resnet50(pretrained=True, norm_layer=FrozenBatchNorm2d,
replace_stride_with_dilation=(False, False, 2))`` behind ``IntermediateLayerGetter`` and a
``Normalizer``.  torchvision, its pretrained weights and ``ml_utils`` are not available on the MI355X
boxes (no network), so the same architecture is written out here with plain torch modules and RANDOM
weights: it exists to put the three custom ops into their real surroundings (a
step-time breakdown), not to reproduce the reference's accuracy.  The convolutions are library
calls (MIOpen) -- outside the hand-written hot path by design.
"""
import re
from typing import Dict, List

import torch
from torch import Tensor, nn

# stage widths / depths of the bottleneck ResNets the reference's BACKBONE_ARCH selects from
_ARCHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


class FrozenBatchNorm2d(nn.Module):
    """Batch norm with fixed statistics and affine parameters: y = x * scale + shift.  All four
    tensors are buffers (nothing to train), as torchvision.ops.misc.FrozenBatchNorm2d."""

    def __init__(self, channels: int, eps: float = 1e-5) -> None:
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(channels))
        self.register_buffer("bias", torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))

    def forward(self, x: Tensor) -> Tensor:
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


class Normalizer(nn.Module):
    """Per-channel input normalisation (ImageNet statistics), what ml_utils' Normalizer does."""

    def __init__(self) -> None:
        super().__init__()
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, x: Tensor) -> Tensor:
        return (x - self.mean) / self.std


class Bottleneck(nn.Module):
    """1x1 reduce, 3x3 (carries the stride and the dilation), 1x1 expand x4, residual."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, dilation: int, project: bool) -> None:
        super().__init__()
        out = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, out, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(out)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, out, 1, stride=stride, bias=False), FrozenBatchNorm2d(out))

    def forward(self, x: Tensor) -> Tensor:
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNetPyramid(nn.Module):
    """conv1 .. layer4 with layer4's stride replaced by dilation 2; returns {"c3", "c4", "c5"} =
    the outputs of layer2, layer3, layer4 (strides 8, 16, 16), like the reference's
    IntermediateLayerGetter(return_layers={"layer2": "c3", "layer3": "c4", "layer4": "c5"})."""

    def __init__(self, depths) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.inplanes, self.dilation = 64, 1
        self.layer1 = self._stage(64, depths[0], stride=1, dilate=False)
        self.layer2 = self._stage(128, depths[1], stride=2, dilate=False)
        self.layer3 = self._stage(256, depths[2], stride=2, dilate=False)
        self.layer4 = self._stage(512, depths[3], stride=2, dilate=True)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if isinstance(m, Bottleneck):
                # random weights + frozen (identity) statistics: damp the residual branches so that 16
                # blocks do not blow the activations up (pretrained statistics do that in the reference)
                m.bn3.weight.fill_(0.25)

    def _stage(self, planes: int, blocks: int, stride: int, dilate: bool) -> nn.Sequential:
        first_dilation = self.dilation
        if dilate:                                   # the stride becomes dilation for the REST of the stage
            self.dilation *= stride
            stride = 1
        layers: List[nn.Module] = [Bottleneck(self.inplanes, planes, stride, first_dilation,
                                              project=stride != 1 or self.inplanes != planes * Bottleneck.expansion)]
        self.inplanes = planes * Bottleneck.expansion
        layers += [Bottleneck(self.inplanes, planes, 1, self.dilation, project=False) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer1(x)
        c3 = self.layer2(x)
        c4 = self.layer3(c3)
        c5 = self.layer4(c4)
        return {"c3": c3, "c4": c4, "c5": c5}


def resnet_backbone(backbone_arch: str, first_trainable_stage: int) -> nn.Module:
    """Backbone: image batch (B, 3, H, W) -> {"c3": (B, 512, H/8, W/8), "c4": (B, 1024, H/16, W/16),
    "c5": (B, 2048, H/16, W/16)}.  Parameters of stages below ``first_trainable_stage`` are frozen
    (reference resnet.py:27-31).  Weights are random (see the module docstring)."""
    if backbone_arch not in _ARCHS:
        raise ValueError(f"unsupported backbone_arch {backbone_arch!r}: one of {sorted(_ARCHS)}")
    body = ResNetPyramid(_ARCHS[backbone_arch])
    body.eval()
    for name, parameter in body.named_parameters():
        match = re.search(r"layer(\d)", name)
        if not (match and int(match.group(1)) >= first_trainable_stage):
            parameter.requires_grad_(False)
    return nn.Sequential(Normalizer(), body)
