"""dev helper: PSROIPool backward time over (R, nT) for the design picked by D2T_PS_BWD (events, C ABI)."""
import os, sys, torch, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT))
from detect_to_track.models import _native
from bench_ops import random_rois, timed, _ws
L = _native.lib
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
k = 7; H, W = 38, 75
mode = os.environ.get("D2T_PS_BWD", "auto")
for R in (300, 1000, 2000, 3000):
    for nT in (4, 8, 16, 31):
        C = nT * 49
        go = [torch.rand(R, nT, k, k, device=dev) for _ in range(4)]
        gin = [torch.empty(C, H, W, device=dev) for _ in range(4)]
        rois = torch.from_numpy(random_rois(R, 1)).to(dev)
        nb = L.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, k, 4); wb = _ws(nb, dev)
        us = timed(lambda i: L.d2t_psroipool_bwd_f32(go[i].data_ptr(), rois.data_ptr(), gin[i].data_ptr(), R, nT, H, W, k, wb.data_ptr(), nb, 0, st), 20, 4)
        print(f"{mode} R={R} nT={nT}: {us:.1f} us", flush=True)
