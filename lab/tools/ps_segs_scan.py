#!/usr/bin/env python3
"""PSROIPool backward, row form: time over the number of RoI ranges per task (scan build; D2T_PS_SEGS is read per call), with a check
against the reference-order kernels first.
    D2T_PS_BWD=rows D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/lib_knobs/libd2t_ops.so python3 lab/tools/ps_segs_scan.py"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
sys.path.insert(0, str(ROOT))
from detect_to_track.models import _native  # noqa: E402
from bench_ops import random_rois, timed, _ws, _check  # noqa: E402

L, dev, k, H, W = _native.lib, "cuda:0", 7, 38, 75
st = torch.cuda.current_stream().cuda_stream
SEGS = [int(x) for x in os.environ.get("SEGS", "1,2,3,4,6,8").split(",")]


def main():
    for R, nT in ((3000, 4), (3000, 31), (3000, 8), (3000, 16), (1000, 4), (1000, 16), (1500, 31), (300, 21), (300, 4)):
        C = nT * 49
        go = [torch.rand(R, nT, k, k, device=dev) for _ in range(4)]
        gin = [torch.empty(C, H, W, device=dev) for _ in range(4)]
        rois = torch.from_numpy(random_rois(R, 1)).to(dev)
        os.environ["D2T_PS_SEGS"] = "8"
        nb = L.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, k, 4)   # the largest the scan uses
        wb = _ws(nb, dev)
        want = torch.empty(C, H, W, device=dev)
        nbg = L.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, k, 4)
        _check(L.d2t_psroipool_bwd_f32(go[0].data_ptr(), rois.data_ptr(), want.data_ptr(), R, nT, H, W, k, wb.data_ptr(), nbg, 1, st))
        row = []
        for s in SEGS:
            os.environ["D2T_PS_SEGS"] = str(s)
            _check(L.d2t_psroipool_bwd_f32(go[0].data_ptr(), rois.data_ptr(), gin[0].data_ptr(), R, nT, H, W, k, wb.data_ptr(), nb, 0, st))
            err = ((gin[0] - want).abs() / (want.abs() + 1.0)).max().item()
            us = timed(lambda i: _check(L.d2t_psroipool_bwd_f32(go[i].data_ptr(), rois.data_ptr(), gin[i].data_ptr(), R, nT, H, W, k, wb.data_ptr(), nb, 0, st)), 20, 4)
            row.append(f"{s}: {us:.1f}" + ("" if err < 1e-5 else f" (ERR {err:.2e})"))
        os.environ["D2T_PS_SEGS"] = "0"
        nb0 = L.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, k, 4)
        us = timed(lambda i: _check(L.d2t_psroipool_bwd_f32(go[i].data_ptr(), rois.data_ptr(), gin[i].data_ptr(), R, nT, H, W, k, wb.data_ptr(), nb0, 0, st)), 20, 4)
        print(f"R={R} nT={nT} [us by ranges]  " + "  ".join(row) + f"  | heuristic: {us:.1f}", flush=True)


if __name__ == "__main__":
    main()
