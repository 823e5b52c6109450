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


@pytest.mark.gpu
@pytest.mark.parametrize("batched", [False, True], ids=["per_pair", "pairs_batched"])
def test_config4_step_at_size_ops_match_reference_kernels(batched, ref_modules, oracle):
    """BASELINE config 4 AT SIZE: one training step of B = 2 frame pairs of 3x608x1008 with 300 regions per frame
    (reference shapes: models/detect_track.py:41-55, cfg/default.yaml:45-50), through detect_to_track/training.py --
    region proposals decoded / filtered / NMS-ed on the device from the live RPN outputs.  Every call the step makes
    into libd2t_ops.so is captured at the `_ext` boundary; shapes are asserted, and each FORWARD call is re-run
    through the reference's own kernels (oracle/_ref) on the captured inputs.  per_pair: the reference's Python loop over
    pairs (B = 1 op calls); pairs_batched: training.py's forward_loss_pairs -- one region-filter call for the 2B frames, one
    fused correlation call with B pairs."""
    from detect_to_track.models import DetectTrackModule, _ext
    from detect_to_track.training import BatchLoader, DataParallelTrainer, RegionProposals, SyntheticPairManager, build_anchors
    ref_corr, ref_roi, ref_ps = ref_modules
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = DetectTrackModule("resnet50", 3, 15, 30, 7, 8, 7).to(dev).train()
    model.c_tracker.fast_forward = True        # as bench_model.py runs it: D2T_IMPL_FAST, the 1024 / 2048-channel levels split channels
    H, W, B, R, T = 608, 1008, 2, 300, 8
    fh, fw = 38, 63
    anchors = build_anchors((fh, fw), [0.001, 0.004, 0.016, 0.064, 0.256], [0.5, 1.0, 2.0])
    assert anchors.shape == (fh * fw * 15, 4)
    params = [p for p in model.parameters() if p.requires_grad]
    trainer = DataParallelTrainer(model, torch.optim.SGD(params, lr=1e-6), torch.tensor([1., 1., 1., 1., 1e-4], device=dev),
                                  RegionProposals(anchors, 0.3, R, 0.5, dev), batched=batched)
    manager = SyntheticPairManager(B, (H, W), len(anchors), R, T, 30, dev, seed=5)

    calls = []
    names = ("pointwise_correlation_levels_forward", "pointwise_correlation_levels_backward", "roipool_forward", "roipool_backward",
             "ps_roipool_forward", "ps_roipool_backward", "region_filter", "region_filter_batched")
    saved = {n: getattr(_ext, n) for n in names}

    def wrap(name):
        def f(*a, **k):
            out = saved[name](*a, **k)
            keep = lambda v: [keep(x) for x in v] if isinstance(v, (list, tuple)) else (v.detach().clone() if torch.is_tensor(v) else v)
            calls.append((name, keep(list(a)), k.get("out", None) is not None, keep(out)))
            return out
        return f
    try:
        for n in names:
            setattr(_ext, n, wrap(n))
        total, _, _ = trainer.train_step(next(iter(BatchLoader(manager, B))))
    finally:
        for n in names:
            setattr(_ext, n, saved[n])
    assert torch.isfinite(total).all()
    for name, p in model.named_parameters():
        assert (p.grad is not None) == p.requires_grad and (p.grad is None or torch.isfinite(p.grad).all()), name
    count = {n: sum(1 for c in calls if c[0] == n) for n in names}
    # per pair: 2 region filters, 2 frames x 2 heads PSROIPool, 1 fused 3-level correlation + 1 ROIPool; backward of each
    nc = 1 if batched else B                                       # fused correlation calls: one for all pairs, or one per pair
    assert count == {"pointwise_correlation_levels_forward": nc, "pointwise_correlation_levels_backward": nc, "roipool_forward": B,
                     "roipool_backward": B, "ps_roipool_forward": 4 * B, "ps_roipool_backward": 4 * B,
                     "region_filter": 0 if batched else 2 * B, "region_filter_batched": 1 if batched else 0}, count
    PB = B if batched else 1                                       # batch size of a fused correlation call
    cells = 289
    for name, a, _, out in calls:
        if name == "region_filter":
            anc, off, conf = a[0], a[1], a[2]
            assert tuple(anc.shape) == (fh * fw * 15, 4) and tuple(out[0].shape) == (R, 4)
            n = int(out[3])
            assert 0 < n <= R and not out[0][n:].any()
        elif name == "region_filter_batched":
            anc, off, conf = a[0], a[1], a[2]
            assert tuple(anc.shape) == (fh * fw * 15, 4) and tuple(off.shape) == (2 * B, fh * fw * 15, 4) and tuple(out[0].shape) == (2 * B, R, 4)
            for f in range(2 * B):
                n = int(out[3][f])
                assert 0 < n <= R and not out[0][f, n:].any()
                single = saved["region_filter"](anc, off[f].contiguous(), conf[f].contiguous(), 0.3, R, 0.5)
                assert torch.equal(single[0], out[0][f]) and int(single[3]) == n
        elif name == "ps_roipool_forward":
            fm, rois, nT, k = a[0], a[1], a[2], a[3]
            assert k == 7 and nT in (31, 4) and tuple(fm.shape) == (nT * 49, fh, fw) and tuple(rois.shape) == (R, 4)
            assert torch.equal(out, ref_ps.ps_roipool_forward(fm, rois, nT, k))                  # bit-exact (ps_roipool_cuda.cu:30-69)
        elif name == "roipool_forward":
            fm, rois, k = a[0], a[1], a[2]
            assert tuple(fm.shape) == (3 * cells + 2 * 512, fh, fw) and tuple(rois.shape) == (T, 4) and k == 7
            torch.testing.assert_close(out, ref_roi.roipool_forward(fm, rois, k), rtol=1e-5, atol=1e-5, equal_nan=True)
        elif name == "pointwise_correlation_levels_forward":
            f0, f1 = a[0], a[1]
            assert [tuple(x.shape) for x in f0] == [(PB, 512, fh, fw), (PB, 1024, fh, fw), (PB, 2048, fh, fw)]   # correlation_tracker.py:57-61
            buf = out
            assert tuple(buf.shape) == (PB, 3 * cells + 2 * 512, fh, fw)
            for l, (x0, x1) in enumerate(zip(f0, f1)):
                want = ref_corr.pointwise_correlation_forward(x0, x1, 8, 1).reshape(PB, fh, fw, cells).permute(0, 3, 1, 2)
                got = buf[:, 2 * 512 + l * cells: 2 * 512 + (l + 1) * cells]
                # the 1024 / 2048-channel levels split their channels over workgroups: f32 rounding of a C-term sum
                torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
        elif name == "ps_roipool_backward":
            assert out.shape[1:] == (fh, fw) and out.shape[0] in (31 * 49, 4 * 49)
        elif name == "roipool_backward":
            assert tuple(out.shape) == (3 * cells + 2 * 512, fh, fw)
