"""One pooling shape, forward + backward, for rocprofv3 --kernel-trace --stats."""
import sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "detect-to-track_amd"))
import numpy as np, torch
from detect_to_track.models import _ext
R, nT, H, W = (int(x) for x in sys.argv[1:5])
rng = np.random.default_rng(0)
rois = torch.from_numpy(np.concatenate([rng.uniform(0.15, 0.85, (R, 2)), rng.uniform(0.05, 0.6, (R, 2))], 1).astype(np.float32)).cuda()
fm = torch.rand(nT * 49, H, W, device="cuda"); go = torch.rand(R, nT, 7, 7, device="cuda")
for _ in range(30):
    _ext.ps_roipool_forward(fm, rois, nT, 7)
    _ext.ps_roipool_backward(go, rois, H, W)
torch.cuda.synchronize()
