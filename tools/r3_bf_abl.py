"""Round-3 development check: time of the bf16x3 backward with parts removed (lab build, D2T_BF_ABL; results are wrong)."""
import sys
sys.path.insert(0, "detect-to-track_amd")
import torch
from detect_to_track.models import _ext
dev = "cuda:0"
B, C, H, W = 8, 256, 38, 63
sets = []
for i in range(6):
    g = torch.Generator().manual_seed(i)
    sets.append((torch.rand(B, H, W, 17, 17, generator=g).to(dev), torch.rand(B, C, H, W, generator=g).to(dev),
                 torch.rand(B, C, H, W, generator=g).to(dev)))
res = []
for rnd in range(4):
    for k in range(6):
        _ext.pointwise_correlation_backward(*sets[k], 8, 1, 4)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(60):
        _ext.pointwise_correlation_backward(*sets[k % 6], 8, 1, 4)
    b.record(); torch.cuda.synchronize()
    res.append(a.elapsed_time(b) / 60 * 1e3)
print("median", round(sorted(res)[len(res) // 2], 1), "min", round(min(res), 1))
