"""dev helper: PSROIPool backward at R=3000 (for rocprofv3 --kernel-trace --stats)."""
import sys, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd")); sys.path.insert(0, str(ROOT))
from detect_to_track.models import _native
from bench_ops import random_rois, timed, _ws
L = _native.lib
dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
k, H, W, R = 7, 38, 63, 300
for nT in (21,):
    go = [torch.rand(R, nT, k, k, device=dev) for _ in range(3)]
    gin = [torch.empty(nT * 49, H, W, device=dev) for _ in range(3)]
    rois = torch.from_numpy(random_rois(R, 1)).to(dev)
    nb = L.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, k, 4); wb = _ws(nb, dev)
    us = timed(lambda i: L.d2t_psroipool_bwd_f32(go[i].data_ptr(), rois.data_ptr(), gin[i].data_ptr(), R, nT, H, W, k, wb.data_ptr(), nb, 0, st), 20, 3)
    print(f"psroipool bwd R={R} nT={nT}: {us:.1f} us")
