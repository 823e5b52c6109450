#!/usr/bin/env python3
"""Band-split forward at the metric shape: time per call with parts of the kernel switched off (stamp build: -DD2T_ENV_KNOBS -DD2T_BAND_STAMPS).
    make -C detect-to-track_amd/csrc -j8 OUT=../lib_stamps EXTRA="-DD2T_ENV_KNOBS -DD2T_BAND_STAMPS"
    D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/lib_stamps/libd2t_ops.so python3 lab/tools/band_ablate.py [cfg,cfg,...] [BxCxHxW]
Ablation bits: 1 no LDS-DMA, 2 no MFMA, 4 no fragment reads from LDS, 8 no stores (results are wrong by design; only the time is read)."""
import ctypes
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _native  # noqa: E402

dev, lib = "cuda:0", _native.lib
lib.d2t_lab_band_dbg.restype, lib.d2t_lab_band_dbg.argtypes = ctypes.c_int, [ctypes.c_int]
CFGS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "141,123").split(",")]
B, C, H, W = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8x256x38x63").split("x"))
ABL = [(0, "all on"), (1, "no DMA"), (2, "no MFMA"), (4, "no fragment reads"), (8, "no stores"), (6, "no MFMA, no fragment reads (DMA + stores)"),
       (9, "no DMA, no stores (LDS reads + MFMA)"), (13, "MFMA only"), (14, "DMA only"), (15, "barriers only")]


def main():
    st = torch.cuda.current_stream().cuda_stream
    nsets = 6
    f0 = [torch.rand(B, C, H, W, device=dev) for _ in range(nsets)]
    f1 = [torch.rand(B, C, H, W, device=dev) for _ in range(nsets)]
    out = [torch.empty(B, H, W, 17, 17, device=dev) for _ in range(nsets)]
    for cfg in CFGS:
        os.environ["D2T_BAND_CFG"] = str(cfg)
        row = []
        for bits, name in ABL:
            assert lib.d2t_lab_band_dbg(bits) == 0
            k = [0]

            def fn():
                i = k[0] % nsets
                k[0] += 1
                assert lib.d2t_corr_fwd_f32(f0[i].data_ptr(), f1[i].data_ptr(), out[i].data_ptr(), B, C, H, W, 8, 1, 0, 0, 0, st) == 0
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30):
                fn()
            b.record()
            torch.cuda.synchronize()
            row.append(f"{name}: {a.elapsed_time(b) / 30 * 1e3:.1f}")
        assert lib.d2t_lab_band_dbg(0) == 0
        print(f"B{B} C{C} {H}x{W} cfg {cfg} [us]  " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
