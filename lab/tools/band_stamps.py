#!/usr/bin/env python3
"""In-kernel clock reads of the band-split forward (scan build, -DD2T_ENV_KNOBS): where a workgroup's time goes.
    D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/lib_knobs/libd2t_ops.so python3 lab/tools/band_stamps.py [cfg,cfg,...]"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _native  # noqa: E402

dev = "cuda:0"
lib = _native.lib
lib.d2t_lab_band_stamps.restype = ctypes.c_int
lib.d2t_lab_band_stamps.argtypes = [ctypes.c_void_p]
CFGS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "41,341,23").split(",")]
SHAPES = [(1, 256, 38, 63), (1, 2048, 38, 75)]
NWG = 8192


def main():
    st = torch.cuda.current_stream().cuda_stream
    for (B, C, H, W) in SHAPES:
        f0, f1 = torch.rand(B, C, H, W, device=dev), torch.rand(B, C, H, W, device=dev)
        out = torch.empty(B, H, W, 17, 17, device=dev)
        for cfg in CFGS:
            os.environ["D2T_BAND_CFG"] = str(cfg)
            stamps = torch.zeros(NWG, 16, dtype=torch.int64, device=dev)
            for it in range(3):
                assert lib.d2t_lab_band_stamps(stamps.data_ptr() if it == 2 else None) == 0
                stamps.zero_()
                torch.cuda.synchronize()
                assert lib.d2t_corr_fwd_f32(f0.data_ptr(), f1.data_ptr(), out.data_ptr(), B, C, H, W, 8, 1, 0, 0, 0, st) == 0
                torch.cuda.synchronize()
            assert lib.d2t_lab_band_stamps(None) == 0
            s = stamps.cpu().numpy().astype(np.int64)
            s = s[s[:, 0] != 0]
            nch = (C + 15) // 16
            med = lambda a: float(np.median(a))
            d = lambda i, j: s[:, j] - s[:, i]
            span_rt = (s[:, 15].max() - s[:, 14].min()) * 10.0 / 1e3
            print(f"B{B} C{C} {H}x{W} cfg {cfg}: {len(s)} active WGs, kernel span {span_rt:.1f} us (first entry -> last exit, 100 MHz clock)")
            print(f"   compute wave 0 [cycles, median / max]: prologue {med(d(0,1)):.0f} / {d(0,1).max()}  loop {med(d(1,2)):.0f} / {d(1,2).max()} "
                  f"(= {med(d(1,2))/nch:.0f} per chunk)  stores {med(d(2,3)):.0f} / {d(2,3).max()}  orphan+drain {med(d(3,4)):.0f} / {d(3,4).max()}  total {med(d(0,4)):.0f} / {d(0,4).max()}")
            wall = (s[:, 15] - s[:, 14]) * 10.0
            clk = np.median(d(0, 4) / np.maximum(wall, 1)) if wall.max() > 0 else 0
            print(f"   in-kernel clock ~{clk:.2f} GHz; WG lifetime median {np.median(wall)/1e3:.1f} us max {wall.max()/1e3:.1f} us; "
                  f"entry skew {(s[:,14].max()-s[:,14].min())*10/1e3:.1f} us")
            print(f"   loader 0: plan+prologue issue {med(d(8,9)):.0f}  wait chunk0 {med(d(9,10)):.0f}  loop {med(d(10,11)):.0f}; per chunk: issue {med(s[:,12])/nch:.0f} wait {med(s[:,13])/nch:.0f}", flush=True)


if __name__ == "__main__":
    main()
