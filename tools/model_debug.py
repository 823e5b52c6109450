"""dev helper: where do non-finite values enter the synthetic model step?"""
import sys, torch, numpy as np
from collections import OrderedDict
sys.path.insert(0, "detect-to-track_amd"); sys.path.insert(0, ".")
from detect_to_track.models import DetectTrackModule
from bench_model import random_rois
torch.manual_seed(0)
m = DetectTrackModule("resnet50", 3, 15, 30, 7, 8, 7).cuda().train()
print("params finite at init:", all(bool(torch.isfinite(p).all()) for p in m.parameters()))
x = torch.rand(2, 3, 608, 1008, device="cuda")
f = m.backbone(x)
print({k: (float(v.abs().max()), bool(torch.isfinite(v).all())) for k, v in f.items()})
o, b, reg = m.rpn(f["c4"]); print("rpn", float(o.abs().max()), float(b.abs().max()), float(reg.abs().max()))
rois = torch.from_numpy(random_rois(300, 1)).cuda()
c, r = m.rcnn(f["c5"][0], rois); print("rcnn", float(c.abs().max()), float(r.abs().max()))
t = m.c_tracker(OrderedDict((k, f[k][0]) for k in f), OrderedDict((k, f[k][1]) for k in f), reg[0], reg[1], rois[:8])
print("tracker", float(t.abs().max()))
for name, out in (("rpn_o", o), ("rpn_b", b), ("rcnn_c", c), ("rcnn_r", r), ("track", t)):
    m.zero_grad()
    out.square().mean().backward(retain_graph=True)
    bad = [n for n, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    gmax = max(float(p.grad.abs().max()) for p in m.parameters() if p.grad is not None)
    print(name, "max |grad|", gmax, "non-finite grads in", bad[:5], len(bad))
