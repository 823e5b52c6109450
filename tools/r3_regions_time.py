"""Round-3 development check: time of one d2t_region_filter_f32 call at the config-4 shape (38 x 63 x 15 anchors)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "detect-to-track_amd"))
import numpy as np
import torch
from detect_to_track.models import _ext
from detect_to_track.training import build_anchors

dev = "cuda:0"
for (h, w, k) in [(38, 63, 300), (38, 63, 16), (38, 63, 128), (38, 63, 512), (38, 63, 3000), (38, 75, 300)]:
    rng = np.random.default_rng(0)
    anchors = torch.from_numpy(build_anchors((h, w), [0.001, 0.004, 0.016, 0.064, 0.256], [0.5, 1.0, 2.0])).to(dev)
    A = anchors.shape[0]
    offs = torch.from_numpy((rng.standard_normal((A, 4)) * 0.2).astype(np.float32)).to(dev)
    conf = torch.softmax(torch.from_numpy(rng.standard_normal((A, 2)).astype(np.float32)), 1)[:, 1].contiguous().to(dev)
    for _ in range(5):
        _ext.region_filter(anchors, offs, conf, 0.05, k, 0.7)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        out = _ext.region_filter(anchors, offs, conf, 0.05, k, 0.7)
    b.record(); torch.cuda.synchronize()
    print(f"A={A} max_dets={k}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call, kept {int(out[3])}")
