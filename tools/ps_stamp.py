import os, sys, torch, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT))
os.environ["D2T_STAMPS"] = "1"
from detect_to_track.models import _native
from bench_ops import random_rois, _ws
L = _native.lib
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
k = 7
for nT, H, W, R in ((4, 38, 75, 300), (31, 38, 75, 3000)):
    C = nT * 49
    go = torch.rand(R, nT, k, k, device=dev); gin = torch.empty(C, H, W, device=dev)
    rois = torch.from_numpy(random_rois(R, 1)).to(dev)
    nb = L.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, k, 4); wb = _ws(nb, dev)
    for i in range(3):
        L.d2t_psroipool_bwd_f32(go.data_ptr(), rois.data_ptr(), gin.data_ptr(), R, nT, H, W, k, wb.data_ptr(), nb, 0, st)
    torch.cuda.synchronize()
    os.environ["D2T_STAMPS_DUMP"] = "1"
    print(f"R={R} nT={nT}", flush=True)
    L.d2t_psroipool_bwd_f32(go.data_ptr(), rois.data_ptr(), gin.data_ptr(), R, nT, H, W, k, wb.data_ptr(), nb, 0, st)
    torch.cuda.synchronize()
    del os.environ["D2T_STAMPS_DUMP"]
