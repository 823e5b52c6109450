"""The rest of the model graph around the three ops (BASELINE config 4): self-contained
ResNet pyramid, RPN, R-FCN, DetectTrackModule -- interfaces of reference models/resnet.py:12-39,
rpn.py:9-52, rfcn.py:10-84, detect_track.py:11-61.  Parity for this row is unpinned (third-party
backbone, random weights); what is tested is the reference's interface contract (names, shapes,
strides, which parameters train) and, on the GPU, that one training step of bench_model.py's form
runs end to end through the HIP ops and reaches every trainable parameter.
"""
import pytest
import torch


def _module():
    from detect_to_track.models import DetectTrackModule
    torch.manual_seed(0)
    return DetectTrackModule("resnet50", 3, 15, 30, 7, 8, 7)      # cfg/default.yaml values


def test_backbone_pyramid_strides_and_frozen_stages():
    m = _module()
    with torch.no_grad():
        f = m.backbone(torch.rand(2, 3, 96, 128))
    assert list(f) == ["c3", "c4", "c5"]                            # detect_track.py / trainer.py:152-233 read these keys
    assert tuple(f["c3"].shape) == (2, 512, 12, 16)                 # stride 8
    assert tuple(f["c4"].shape) == (2, 1024, 6, 8)                  # stride 16
    assert tuple(f["c5"].shape) == (2, 2048, 6, 8)                  # stride 16: layer4's stride replaced by dilation
    dil = {mod.dilation[0] for mod in m.backbone[1].layer4.modules() if isinstance(mod, torch.nn.Conv2d) and mod.kernel_size == (3, 3)}
    assert dil == {1, 2}                                            # first block keeps dilation 1, the rest 2 (torchvision's rule)
    for name, p in m.backbone.named_parameters():                   # resnet.py:27-31, first_trainable_stage = 3
        stage = int(name.split("layer")[1][0]) if "layer" in name else 0
        assert p.requires_grad == (stage >= 3), name
    assert not any(isinstance(mod, torch.nn.BatchNorm2d) for mod in m.modules())   # FrozenBatchNorm2d only: no batch statistics
    with pytest.raises(ValueError):
        from detect_to_track.models import resnet_backbone
        resnet_backbone("vgg16", 3)


def test_rpn_and_module_contract():
    m = _module()
    assert (m.stage3_outchannels, m.stage4_outchannels, m.stage5_outchannels) == (512, 1024, 2048)
    assert m.rpn.conv.out_channels == 512 and m.c_tracker.fc_channels == (3 * 17 * 17 + 2 * 512) * 49
    x = torch.rand(2, 1024, 5, 7)
    o_hat, b_hat, feats = m.rpn(x)
    assert tuple(o_hat.shape) == (2, 5 * 7 * 15, 2) and tuple(b_hat.shape) == (2, 5 * 7 * 15, 4) and tuple(feats.shape) == (2, 512, 5, 7)
    torch.testing.assert_close(o_hat.sum(-1), torch.ones(2, 5 * 7 * 15))           # softmax over object / not object
    # anchors of one cell stay together (rpn.py:25-31): entry (h*W + w)*15 + a  <-  channel a*4 + k at (h, w)
    raw = m.rpn.reg_fc(torch.relu(m.rpn.conv(x)))
    assert torch.equal(b_hat[1, (3 * 7 + 2) * 15 + 4], raw[1, 16:20, 3, 2])
    with pytest.raises(NotImplementedError):
        m()


@pytest.mark.gpu
def test_training_step_reaches_every_trainable_parameter():
    from collections import OrderedDict
    from conftest import random_rois
    m = _module().cuda().train()
    x = torch.rand(2, 3, 160, 208, device="cuda")                  # c4 10x13 -> generic correlation kernels; 320x336 below: MFMA
    rois = torch.from_numpy(random_rois(24, 1)).cuda()
    for shape in ((160, 208), (320, 336)):
        m.zero_grad()
        x = torch.rand(2, 3, *shape, device="cuda")
        f = m.backbone(x)
        o_hat, b_hat, reg = m.rpn(f["c4"])
        c_hat, r_hat = m.rcnn(f["c5"][0], rois)
        assert tuple(c_hat.shape) == (24, 31) and tuple(r_hat.shape) == (24, 4)
        torch.testing.assert_close(c_hat.sum(1), torch.ones(24, device="cuda"))
        t_hat = m.c_tracker(OrderedDict((k, f[k][0]) for k in f), OrderedDict((k, f[k][1]) for k in f), reg[0], reg[1], rois[:5])
        assert tuple(t_hat.shape) == (5, 4)
        (o_hat.square().mean() + b_hat.square().mean() + c_hat.square().mean() + r_hat.square().mean() + t_hat.square().mean()).backward()
        for name, p in m.named_parameters():
            assert (p.grad is not None) == p.requires_grad, name
            if p.grad is not None:
                assert torch.isfinite(p.grad).all(), name
