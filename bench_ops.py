#!/usr/bin/env python3
"""bench_ops.py -- secondary benchmark: every op of the hot path at the shapes BASELINE.json and
SURVEY.md section 8d name, one JSON line per (op, shape, direction).  Not the driver's contract
(that is bench.py); used for DESIGN.md's tables and the rocprof summaries under profiles/.

    python bench_ops.py [--iters 50] [--impl 0|1]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _ext, _native  # noqa: E402

L = _native.lib


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)


def _check(rc):
    if rc:
        raise RuntimeError(_native.error_string(rc).decode())

HBM = 8000.0


def random_rois(R, seed):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(0.15, 0.85, (R, 2)), rng.uniform(0.05, 0.6, (R, 2))], 1).astype(np.float32)


def timed(fn, iters, nsets):
    for i in range(3):
        fn(i % nsets)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % nsets)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3           # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--impl", type=int, default=0)
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    out = []

    def emit(op, shape, direction, us, nbytes, extra=None):
        line = dict(op=op, shape=shape, dir=direction, us=round(us, 2), algo_MB=round(nbytes / 1e6, 3),
                    GBps=round(nbytes / us / 1e3, 1), pct_hbm=round(100 * nbytes / us / 1e3 / HBM, 2), impl=args.impl)
        if extra:
            line.update(extra)
        print(json.dumps(line), flush=True)

    # ---- pooling, config 3 and the model-true shapes.  Timed through the C ABI with
    # caller-owned outputs and workspace (what the autograd Function does, minus the allocator).
    st = torch.cuda.current_stream().cuda_stream
    k = 7
    for name, C, H, W, R in (("roipool", 1024, 38, 63, 300), ("roipool", 1891, 38, 75, 8)):
        nsets = min(8, max(2, int(600e6 // (R * C * k * k * 8 + C * H * W * 8)) + 1))
        fm = [torch.rand(C, H, W, device=dev) for _ in range(nsets)]
        go = [torch.rand(R, C, k, k, device=dev) for _ in range(nsets)]
        out = [torch.empty(R, C, k, k, device=dev) for _ in range(nsets)]
        gin = [torch.empty(C, H, W, device=dev) for _ in range(nsets)]
        rois = torch.from_numpy(random_rois(R, 0)).to(dev)
        nf, nbw = L.d2t_roipool_fwd_workspace_bytes(R, C, H, W, k, 4), L.d2t_roipool_bwd_workspace_bytes(R, C, H, W, k, 4)
        wf, wb = _ws(nf, dev), _ws(nbw, dev)
        nb = R * C * k * k * 4 + C * H * W * 4 + R * 16
        emit(name, f"R{R}_C{C}_{H}x{W}_k{k}", "fwd", timed(lambda i: _check(L.d2t_roipool_fwd_f32(
            fm[i].data_ptr(), rois.data_ptr(), out[i].data_ptr(), R, C, H, W, k, wf.data_ptr(), nf, args.impl, st)), args.iters, nsets), nb)
        emit(name, f"R{R}_C{C}_{H}x{W}_k{k}", "bwd", timed(lambda i: _check(L.d2t_roipool_bwd_f32(
            go[i].data_ptr(), rois.data_ptr(), gin[i].data_ptr(), R, C, H, W, k, wb.data_ptr(), nbw, args.impl, st)), args.iters, nsets), nb)
        del fm, go, out, gin
    for nT, H, W, R in ((21, 38, 63, 300), (31, 38, 75, 300), (4, 38, 75, 300), (31, 38, 75, 3000), (4, 38, 75, 3000)):
        C = nT * k * k
        fm = [torch.rand(C, H, W, device=dev) for _ in range(4)]
        go = [torch.rand(R, nT, k, k, device=dev) for _ in range(4)]
        out = [torch.empty(R, nT, k, k, device=dev) for _ in range(4)]
        gin = [torch.empty(C, H, W, device=dev) for _ in range(4)]
        rois = torch.from_numpy(random_rois(R, 1)).to(dev)
        nbw = L.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, k, 4)
        wb = _ws(nbw, dev)
        nfw = L.d2t_psroipool_fwd_workspace_bytes(R, nT, H, W, k, 4)
        wf = _ws(nfw, dev)
        nb = R * nT * k * k * 4 + C * H * W * 4 + R * 16
        emit("psroipool", f"R{R}_nT{nT}_{H}x{W}_k{k}", "fwd", timed(lambda i: _check(L.d2t_psroipool_fwd_f32(
            fm[i].data_ptr(), rois.data_ptr(), out[i].data_ptr(), R, nT, H, W, k, wf.data_ptr(), nfw, args.impl, st)), args.iters, 4), nb)
        emit("psroipool", f"R{R}_nT{nT}_{H}x{W}_k{k}", "bwd", timed(lambda i: _check(L.d2t_psroipool_bwd_f32(
            go[i].data_ptr(), rois.data_ptr(), gin[i].data_ptr(), R, nT, H, W, k, wb.data_ptr(), nbw, args.impl, st)), args.iters, 4), nb)

    # ---- tracker glue (correlation_tracker.py:64-83): three correlations + permute + cat + ROIPool
    #      as the reference composes them, against the fused levels call writing into the concat buffer
    H, W, cr, R = 38, 75, 512, 8
    Cs = (512, 1024, 2048)
    f0 = [torch.rand(1, C, H, W, device=dev) for C in Cs]
    f1 = [torch.rand(1, C, H, W, device=dev) for C in Cs]
    reg0, reg1 = torch.rand(cr, H, W, device=dev), torch.rand(cr, H, W, device=dev)
    rois_t = torch.from_numpy(random_rois(R, 2)).to(dev)

    def unfused(_):
        feats = []
        for a, b in zip(f0, f1):
            cf = _ext.pointwise_correlation_forward(a, b, 8, 1, args.impl)
            feats.append(cf.squeeze(0).view(H, W, -1).permute(2, 0, 1))
        return _ext.roipool_forward(torch.cat([reg0, reg1, *feats]), rois_t, 7, args.impl)

    buf = torch.empty(1, 2 * cr + 3 * 289, H, W, device=dev)

    def fused(_):
        buf[0, :cr] = reg0
        buf[0, cr:2 * cr] = reg1
        _ext.pointwise_correlation_levels_forward(f0, f1, 8, 1, out=(buf, 2 * cr), impl=args.impl)
        return _ext.roipool_forward(buf[0], rois_t, 7, args.impl)

    torch.testing.assert_close(unfused(0), fused(0), rtol=1e-5, atol=1e-4)   # the 1024 / 2048-channel levels split channels: f32 rounding
    nb = sum(2 * C * H * W * 4 for C in Cs) + 3 * 289 * H * W * 4
    emit("tracker_fwd", "3corr+cat+roipool_R8_38x75", "unfused", timed(unfused, args.iters, 1), nb)
    emit("tracker_fwd", "3corr+cat+roipool_R8_38x75", "fused", timed(fused, args.iters, 1), nb)
    del f0, f1, buf

    # ---- correlation: metric shape, config 2, model-true shapes
    for B, C, H, W in ((8, 256, 38, 63), (1, 256, 38, 63), (1, 512, 38, 75), (1, 1024, 38, 75), (1, 2048, 38, 75)):
        d = 8
        nsets = 6 if B > 1 else 8
        f0 = [torch.rand(B, C, H, W, device=dev) for _ in range(nsets)]
        f1 = [torch.rand(B, C, H, W, device=dev) for _ in range(nsets)]
        go = [torch.rand(B, H, W, 17, 17, device=dev) for _ in range(nsets)]
        inb, outb = B * C * H * W * 4, B * H * W * 289 * 4
        vox = B * H * W * 289
        out = [torch.empty(B, H, W, 17, 17, device=dev) for _ in range(nsets)]
        g0 = [torch.empty(B, C, H, W, device=dev) for _ in range(nsets)]
        g1 = [torch.empty(B, C, H, W, device=dev) for _ in range(nsets)]
        nws = L.d2t_corr_fwd_workspace_bytes(B, C, H, W, d, 1, 4)       # > 0: the call splits channels over workgroups
        wsf = torch.empty(max(nws, 1), dtype=torch.uint8, device=dev)
        tf = timed(lambda i: _check(L.d2t_corr_fwd_f32(f0[i].data_ptr(), f1[i].data_ptr(), out[i].data_ptr(),
                                                        B, C, H, W, d, 1, wsf.data_ptr() if nws else 0, nws, args.impl, st)), args.iters, nsets)
        tb = timed(lambda i: _check(L.d2t_corr_bwd_f32(go[i].data_ptr(), f0[i].data_ptr(), f1[i].data_ptr(),
                                                        g0[i].data_ptr(), g1[i].data_ptr(),
                                                        B, C, H, W, d, 1, 0, 0, args.impl, st)), args.iters, nsets)
        emit("corr", f"B{B}_C{C}_{H}x{W}_d8", "fwd", tf, 2 * inb + outb, dict(gvox_s=round(vox / tf / 1e3, 2)))
        emit("corr", f"B{B}_C{C}_{H}x{W}_d8", "bwd", tb, outb + 4 * inb, dict(gvox_s=round(vox / tb / 1e3, 2)))
        del f0, f1, go, out, g0, g1


if __name__ == "__main__":
    main()
