"""dev helper: ROIPool forward time versus the number of RoIs (tuned kernel demanded)."""
import sys, torch, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT))
from detect_to_track.models import _native
from bench_ops import random_rois, timed, _ws
L = _native.lib
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
C, H, W, k = 1024, 38, 63, 7
for R in (32, 75, 150, 300, 600, 1200):
    nsets = 4
    fm = [torch.rand(C, H, W, device=dev) for _ in range(nsets)]
    out = [torch.empty(R, C, k, k, device=dev) for _ in range(nsets)]
    rois = torch.from_numpy(random_rois(R, 0)).to(dev)
    nf = L.d2t_roipool_fwd_workspace_bytes(R, C, H, W, k, 4); wf = _ws(nf, dev)
    us = timed(lambda i: L.d2t_roipool_fwd_f32(fm[i].data_ptr(), rois.data_ptr(), out[i].data_ptr(), R, C, H, W, k, wf.data_ptr(), nf, 2, st), 30, nsets)
    print(R, round(us, 1), "us", round(R * C * 49 * 4 / us / 1e3), "GB/s out")
