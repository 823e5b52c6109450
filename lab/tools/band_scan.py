#!/usr/bin/env python3
"""Scan of the band-split forward (csrc/d2t_corr_fwd_band.hip) over workgroup shapes: bit-exactness against the generic kernel
(both layouts) and time per call, next to the default dispatch of the product library.  Needs a knob build:
    make -C detect-to-track_amd/csrc -j8 OUT=../lib_knobs EXTRA=-DD2T_ENV_KNOBS
    gpurun -- 'D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/lib_knobs/libd2t_ops.so python3 lab/tools/band_scan.py'
(D2T_BAND_CFG is read per call by that build: 0 = the segment kernels, 41 = TW 4 x NB 1, ...)."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _ext, _native  # noqa: E402

dev = "cuda:0"
CFGS = [int(x) for x in os.environ.get("SCAN_CFGS", "0,41,42,43,23,22,21,11,16").split(",")]
SHAPES = [tuple(int(v) for v in x.split("x")) for x in os.environ["SCAN_SHAPES"].split(",")] if "SCAN_SHAPES" in os.environ else [(1, 256, 38, 63), (1, 512, 38, 75), (1, 1024, 38, 75), (1, 2048, 38, 75), (2, 256, 38, 63), (2, 1024, 38, 75)]
CHECK = [(1, 40, 38, 63), (2, 24, 21, 24), (1, 17, 5, 20), (1, 33, 38, 75), (3, 16, 13, 41)]


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    torch.manual_seed(0)
    bad = 0
    for cfg in CFGS:
        os.environ["D2T_BAND_CFG"] = str(cfg)
        for (B, C, H, W) in CHECK:
            a, b = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
            ref = _ext.pointwise_correlation_forward(a, b, 8, 1, _native.IMPL_GENERIC)
            got = _ext.pointwise_correlation_forward(a, b, 8, 1, _native.IMPL_AUTO)
            ok = torch.equal(ref, got)
            buf = torch.full((B, 289 + 7, H, W), float("nan"), device=dev)
            _ext.pointwise_correlation_levels_forward([a], [b], 8, 1, out=(buf, 3), impl=_native.IMPL_AUTO)
            cm = buf[:, 3:292].reshape(B, 17, 17, H, W).permute(0, 3, 4, 1, 2)
            ok2 = torch.equal(ref, cm.contiguous()) and bool(torch.isnan(buf[:, :3]).all()) and bool(torch.isnan(buf[:, 292:]).all())
            if not (ok and ok2):
                bad += 1
                d = (ref - got)
                print(f"cfg {cfg} shape {(B, C, H, W)}: MISMATCH ref-layout {ok} channel-major {ok2}; "
                      f"bad cells {(ref != got).sum().item()} of {ref.numel()}, nan {torch.isnan(got).sum().item()}", flush=True)
        print(f"cfg {cfg}: checks done, mismatches so far {bad}", flush=True)
    for (B, C, H, W) in SHAPES:
        nsets = 6
        f0 = [torch.rand(B, C, H, W, device=dev) for _ in range(nsets)]
        f1 = [torch.rand(B, C, H, W, device=dev) for _ in range(nsets)]
        out = [torch.empty(B, H, W, 17, 17, device=dev) for _ in range(nsets)]
        st = torch.cuda.current_stream().cuda_stream
        row = []
        for cfg in CFGS:
            os.environ["D2T_BAND_CFG"] = str(cfg)
            k = [0]

            def fn():
                i = k[0] % nsets
                k[0] += 1
                rc = _native.lib.d2t_corr_fwd_f32(f0[i].data_ptr(), f1[i].data_ptr(), out[i].data_ptr(), B, C, H, W, 8, 1, 0, 0, 0, st)
                assert rc == 0, rc
            row.append((cfg, round(timed(fn), 1)))
        print(f"B{B} C{C} {H}x{W}: " + "  ".join(f"{c}:{t}" for c, t in row), flush=True)
    print("MISMATCHES", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
