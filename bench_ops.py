#!/usr/bin/env python3
"""bench_ops.py -- secondary benchmark: every op of the hot path at the shapes BASELINE.json and
SURVEY.md section 8d name, one JSON line per (op, shape, direction).  Not the driver's contract
(that is bench.py) -- but bench.py calls `measure()` after its timed region and carries the
result as `ops[]` in its line, so the driver's record holds these numbers too.

    python bench_ops.py [--iters 50] [--impl 0|1] [--full 1]

Every op is called through the C ABI (caller-owned outputs and workspace: what the autograd
Function does, minus the allocator) on torch's current stream, with enough rotated buffer sets
that a set's bytes are not found in the 256 MiB Infinity Cache again.  Roofs: the correlation is
f32-matrix-bound (157.3 TF/s), the pooling ops are byte movers (8 TB/s HBM); reference shapes:
correlation_tracker.py:57-70,82, rfcn.py:40, cfg/default.yaml:9,22,45-50.
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _ext, _native  # noqa: E402

L = _native.lib
HBM_GBS = 8000.0
F32_TF = 157.3
K = 7

# rocprofv3-measured HBM bytes per call of every (op, shape, direction) below: profiles/traffic.json "ops", written by tools/ops_pmc.sh
# (separate --pmc FETCH_SIZE / WRITE_SIZE passes of THIS script; bytes = (2 FETCH_SIZE + WRITE_SIZE) KB, FETCH doubled per the MI355X guide)
try:
    TRAFFIC = json.loads((ROOT / "profiles" / "traffic.json").read_text()).get("ops", {})
except Exception:
    TRAFFIC = {}
MARKER = None          # tools/ops_pmc.sh: called with (label, calls) in front of every timed series and with (None, 1) behind it


def label_of(op, shape, direction):
    return f"{op}/{shape}/{direction}"


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)


def _check(rc):
    if rc:
        raise RuntimeError(_native.error_string(rc).decode())


def random_rois(R, seed):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(0.15, 0.85, (R, 2)), rng.uniform(0.05, 0.6, (R, 2))], 1).astype(np.float32)


def timed(fn, iters, nsets, label=None):
    if MARKER is not None:
        MARKER(label, iters + 3)
    for i in range(3):
        fn(i % nsets)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % nsets)
    b.record()
    torch.cuda.synchronize()
    if MARKER is not None:
        MARKER(None, 1)                              # closes the series: what follows (the next shape's inputs) belongs to no label
    return a.elapsed_time(b) / iters * 1e3           # us


def _entry(op, shape, direction, us, nbytes, flops, impl, note=None):
    """One ops[] entry: us per call (HIP events around `iters` back-to-back calls), algorithmic MB / GFLOP, and the
    fraction of the op's own roof (correlation: f32 matrix peak; pooling: HBM)."""
    e = dict(op=op, shape=shape, dir=direction, us=round(us, 2), algo_MB=round(nbytes / 1e6, 3),
             GBps=round(nbytes / us / 1e3, 1), pct_hbm=round(100 * nbytes / us / 1e3 / HBM_GBS, 2), impl=impl)
    if flops:
        e.update(GFLOP=round(flops / 1e9, 4), TFLOPs=round(flops / us / 1e6, 2), roof="f32_matrix",
                 frac=round(flops / us / 1e6 / F32_TF, 4))
    else:
        e.update(roof="hbm", frac=round(nbytes / us / 1e3 / HBM_GBS, 4))
    if note:
        e["note"] = note
    t = TRAFFIC.get(label_of(op, shape, direction))
    if t:                                            # rocprof-reported HBM bytes of the same call (north_star: achieved HBM GB/s)
        e["traffic"] = int(t["hbm_bytes"])
        e["traffic_over_algorithmic"] = round(t["hbm_bytes"] / nbytes, 3)
        e["hbm_GBps_measured"] = round(t["hbm_bytes"] / us / 1e3, 1)
    return e


def corr_flops(B, C, H, W, d=8):
    """2 flops per MAC over the window cells the reference's loops visit (pointwise_correlation_cuda.cu:92-107)."""
    ni = sum(min(i + d, H) - max(0, i - d) for i in range(H))
    nj = sum(min(j + d, W) - max(0, j - d) for j in range(W))
    return 2 * B * C * ni * nj


def measure_roipool(dev, R, C, H, W, impl, iters, st):
    per_set = R * C * K * K * 8 + C * H * W * 8
    nsets = min(8, max(2, int(600e6 // per_set) + 1))
    fm = [torch.rand(C, H, W, device=dev) for _ in range(nsets)]
    go = [torch.rand(R, C, K, K, device=dev) for _ in range(nsets)]
    out = [torch.empty(R, C, K, K, device=dev) for _ in range(nsets)]
    gin = [torch.empty(C, H, W, device=dev) for _ in range(nsets)]
    rois = torch.from_numpy(random_rois(R, 0)).to(dev)
    nf, nbw = L.d2t_roipool_fwd_workspace_bytes(R, C, H, W, K, 4), L.d2t_roipool_bwd_workspace_bytes(R, C, H, W, K, 4)
    wf, wb = _ws(nf, dev), _ws(nbw, dev)
    nb = R * C * K * K * 4 + C * H * W * 4 + R * 16
    shape = f"R{R}_C{C}_{H}x{W}_k{K}"
    tf = timed(lambda i: _check(L.d2t_roipool_fwd_f32(fm[i].data_ptr(), rois.data_ptr(), out[i].data_ptr(), R, C, H, W, K,
                                                      wf.data_ptr(), nf, impl, st)), iters, nsets, label_of("roipool", shape, "fwd"))
    tb = timed(lambda i: _check(L.d2t_roipool_bwd_f32(go[i].data_ptr(), rois.data_ptr(), gin[i].data_ptr(), R, C, H, W, K,
                                                      wb.data_ptr(), nbw, impl, st)), iters, nsets, label_of("roipool", shape, "bwd"))
    return [_entry("roipool", shape, "fwd", tf, nb, 0, impl), _entry("roipool", shape, "bwd", tb, nb, 0, impl)]


def measure_psroipool(dev, R, nT, H, W, impl, iters, st):
    C = nT * K * K
    nsets = 4
    fm = [torch.rand(C, H, W, device=dev) for _ in range(nsets)]
    go = [torch.rand(R, nT, K, K, device=dev) for _ in range(nsets)]
    out = [torch.empty(R, nT, K, K, device=dev) for _ in range(nsets)]
    gin = [torch.empty(C, H, W, device=dev) for _ in range(nsets)]
    rois = torch.from_numpy(random_rois(R, 1)).to(dev)
    nbw = L.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, K, 4)
    nfw = L.d2t_psroipool_fwd_workspace_bytes(R, nT, H, W, K, 4)
    wb, wf = _ws(nbw, dev), _ws(nfw, dev)
    nb = R * nT * K * K * 4 + C * H * W * 4 + R * 16
    shape = f"R{R}_nT{nT}_{H}x{W}_k{K}"
    tf = timed(lambda i: _check(L.d2t_psroipool_fwd_f32(fm[i].data_ptr(), rois.data_ptr(), out[i].data_ptr(), R, nT, H, W, K,
                                                        wf.data_ptr(), nfw, impl, st)), iters, nsets, label_of("psroipool", shape, "fwd"))
    tb = timed(lambda i: _check(L.d2t_psroipool_bwd_f32(go[i].data_ptr(), rois.data_ptr(), gin[i].data_ptr(), R, nT, H, W, K,
                                                        wb.data_ptr(), nbw, impl, st)), iters, nsets, label_of("psroipool", shape, "bwd"))
    note = "launch/latency-bound: a few MB per call (SURVEY 8d)"
    return [_entry("psroipool", shape, "fwd", tf, nb, 0, impl, note), _entry("psroipool", shape, "bwd", tb, nb, 0, impl, note)]


def measure_corr(dev, B, C, H, W, impl, iters, st):
    d = 8
    inb, outb = B * C * H * W * 4, B * H * W * 289 * 4
    nsets = max(2, min(8, int(600e6 // (4 * inb + 2 * outb)) + 1))
    f0 = [torch.rand(B, C, H, W, device=dev) for _ in range(nsets)]
    f1 = [torch.rand(B, C, H, W, device=dev) for _ in range(nsets)]
    go = [torch.rand(B, H, W, 17, 17, device=dev) for _ in range(nsets)]
    out = [torch.empty(B, H, W, 17, 17, device=dev) for _ in range(nsets)]
    g0 = [torch.empty(B, C, H, W, device=dev) for _ in range(nsets)]
    g1 = [torch.empty(B, C, H, W, device=dev) for _ in range(nsets)]
    nws = L.d2t_corr_fwd_workspace_bytes(B, C, H, W, d, 1, 4)       # > 0: the call MAY split channels over workgroups
    wsf = _ws(nws, dev)
    fl = corr_flops(B, C, H, W)
    shape = f"B{B}_C{C}_{H}x{W}_d8"
    vox = B * H * W * 289
    tf = timed(lambda i: _check(L.d2t_corr_fwd_f32(f0[i].data_ptr(), f1[i].data_ptr(), out[i].data_ptr(),
                                                    B, C, H, W, d, 1, wsf.data_ptr() if nws else 0, nws, impl, st)), iters, nsets, label_of("corr", shape, "fwd"))
    tb = timed(lambda i: _check(L.d2t_corr_bwd_f32(go[i].data_ptr(), f0[i].data_ptr(), f1[i].data_ptr(),
                                                    g0[i].data_ptr(), g1[i].data_ptr(),
                                                    B, C, H, W, d, 1, 0, 0, impl, st)), iters, nsets, label_of("corr", shape, "bwd"))
    ef = _entry("corr", shape, "fwd", tf, 2 * inb + outb, fl, impl)
    eb = _entry("corr", shape, "bwd", tb, outb + 4 * inb, 2 * fl, impl)
    ef["gvox_s"], eb["gvox_s"] = round(vox / tf / 1e3, 2), round(vox / tb / 1e3, 2)
    res = [ef, eb]
    if nws and impl == _native.IMPL_AUTO:
        # the opt-in forward (D2T_IMPL_FAST): channels of a small grid split over workgroups, within 1e-5, not bit-identical
        tq = timed(lambda i: _check(L.d2t_corr_fwd_f32(f0[i].data_ptr(), f1[i].data_ptr(), out[i].data_ptr(),
                                                        B, C, H, W, d, 1, wsf.data_ptr(), nws, _native.IMPL_FAST, st)), iters, nsets,
                   label_of("corr", shape, "fwd_fast"))
        eq = _entry("corr", shape, "fwd_fast", tq, 2 * inb + outb, fl, _native.IMPL_FAST,
                    "opt-in D2T_IMPL_FAST: channel split, deterministic, within 1e-5 of the reference")
        eq["gvox_s"] = round(vox / tq / 1e3, 2)
        res.insert(1, eq)
    return res


def measure_tracker(dev, impl, iters):
    """Tracker glue (correlation_tracker.py:64-83): three correlations + permute + cat + ROIPool as the
    reference composes them, against the fused levels call writing into the concat buffer."""
    H, W, cr, R = 38, 75, 512, 8
    Cs = (512, 1024, 2048)
    f0 = [torch.rand(1, C, H, W, device=dev) for C in Cs]
    f1 = [torch.rand(1, C, H, W, device=dev) for C in Cs]
    reg0, reg1 = torch.rand(cr, H, W, device=dev), torch.rand(cr, H, W, device=dev)
    rois_t = torch.from_numpy(random_rois(R, 2)).to(dev)

    def unfused(_):
        feats = []
        for a, b in zip(f0, f1):
            cf = _ext.pointwise_correlation_forward(a, b, 8, 1, impl)
            feats.append(cf.squeeze(0).view(H, W, -1).permute(2, 0, 1))
        return _ext.roipool_forward(torch.cat([reg0, reg1, *feats]), rois_t, 7, impl)

    buf = torch.empty(1, 2 * cr + 3 * 289, H, W, device=dev)

    def fused(_, corr_impl=impl):
        buf[0, :cr] = reg0
        buf[0, cr:2 * cr] = reg1
        _ext.pointwise_correlation_levels_forward(f0, f1, 8, 1, out=(buf, 2 * cr), impl=corr_impl)
        return _ext.roipool_forward(buf[0], rois_t, 7, impl)

    assert torch.equal(unfused(0), fused(0)) or impl != _native.IMPL_AUTO
    torch.testing.assert_close(unfused(0), fused(0, _native.IMPL_FAST), rtol=1e-5, atol=1e-4)   # the 1024 / 2048-channel levels split channels
    nb = sum(2 * C * H * W * 4 for C in Cs) + 3 * 289 * H * W * 4
    fl = sum(corr_flops(1, C, H, W) for C in Cs)
    shape = "3corr+cat+roipool_R8_38x75"
    return [_entry("tracker_fwd", shape, "unfused", timed(unfused, iters, 1, label_of("tracker_fwd", shape, "unfused")), nb, fl, impl),
            _entry("tracker_fwd", shape, "fused", timed(fused, iters, 1, label_of("tracker_fwd", shape, "fused")), nb, fl, impl),
            _entry("tracker_fwd", shape, "fused_fast", timed(lambda i: fused(i, _native.IMPL_FAST), iters, 1, label_of("tracker_fwd", shape, "fused_fast")),
                   nb, fl, _native.IMPL_FAST)]


def measure(dev="cuda:0", impl=0, iters=20, full=False):
    """The ops[] list of bench.py's line (full=False: about 3 s on an MI355X) / every line of this script (full=True).
    Order: config 2, the model's three correlations, ROIPool config 3 + eval path + tracker, PSROIPool config 3 + the
    model's shapes (rfcn.py:40: nT = 31 classification, 4 regression; 300 regions at eval / config 3, 3000 in training)."""
    dev = torch.device(dev)
    torch.manual_seed(0)
    st = torch.cuda.current_stream(dev).cuda_stream
    ops = []
    with torch.cuda.device(dev):
        for B, C, H, W in ((1, 256, 38, 63), (1, 512, 38, 75), (1, 1024, 38, 75), (1, 2048, 38, 75)) + (((8, 256, 38, 63),) if full else ()):
            ops += measure_corr(dev, B, C, H, W, impl, iters, st)
        for C, H, W, R in ((1024, 38, 63, 300), (1891, 38, 75, 300), (1891, 38, 75, 8)):
            ops += measure_roipool(dev, R, C, H, W, impl, iters, st)
        for nT, H, W, R in ((21, 38, 63, 300), (31, 38, 75, 3000), (4, 38, 75, 3000)) + (((31, 38, 75, 300), (4, 38, 75, 300)) if full else ()):
            ops += measure_psroipool(dev, R, nT, H, W, impl, iters, st)
        if full:
            ops += measure_tracker(dev, impl, iters)
    return ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--impl", type=int, default=0)
    ap.add_argument("--full", type=int, default=1, help="1: also the metric shape, the small PSROIPool shapes and the tracker glue")
    ap.add_argument("--pmc-markers", default=None, help="tools/ops_pmc.sh: write (label, calls) of every timed series to this JSON-lines file and "
                    "launch a marker (two 1 x 1 k_corr_mask dispatches) in front of each, so that a rocprofv3 --pmc pass can be cut into the series")
    args = ap.parse_args()
    if args.pmc_markers:
        global MARKER
        mk = torch.empty(64, dtype=torch.uint8, device="cuda:0")
        fh = open(args.pmc_markers, "w")

        def MARKER(label, calls):
            fh.write(json.dumps({"label": label, "calls": calls}) + "\n")
            fh.flush()
            for _ in range(2):                                      # the marker is a PAIR of 1 x 1 masks (tools/ops_pmc_reduce.py): no series launches that
                _check(L.d2t_corr_mask(mk.data_ptr(), 1, 1, 0, 1, torch.cuda.current_stream().cuda_stream))
    for e in measure("cuda:0", args.impl, args.iters, bool(args.full)):
        print(json.dumps(e), flush=True)


if __name__ == "__main__":
    main()
