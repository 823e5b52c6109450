#!/usr/bin/env python3
"""In-kernel clock reads (knob build, -DD2T_ENV_KNOBS) of the pooling kernels: where a workgroup's time goes.
    D2T_OPS_LIBRARY=$PWD/detect-to-track_amd/lib_knobs/libd2t_ops.so python3 lab/tools/kstamps.py roipool_fwd"""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _native  # noqa: E402
import bench_ops  # noqa: E402

lib, dev = _native.lib, "cuda:0"
NWG = 16384


def run(setter, call, names, extra=None):
    fn = getattr(lib, setter)
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p]
    stamps = torch.zeros(NWG, 16, dtype=torch.int64, device=dev)
    for it in range(3):
        assert fn(stamps.data_ptr() if it == 2 else None) == 0
        stamps.zero_()
        torch.cuda.synchronize()
        call()
        torch.cuda.synchronize()
    assert fn(None) == 0
    s = stamps.cpu().numpy()
    s = s[s[:, 0] != 0]
    print(f"{len(s)} workgroups stamped")
    for i in range(1, len(names)):
        on = (s[:, i] != 0) & (s[:, i - 1] != 0)
        if on.any():
            d = s[on, i] - s[on, i - 1]
            print(f"   {names[i - 1]:28s} -> {names[i]:28s}: median {np.median(d):8.0f}  max {d.max():8d}  ({on.sum()} WGs)")
    if s[:, 14].any() and s[:, 15].any():                            # 100 MHz wall clock at entry / exit: how many workgroups run at a time
        t0, t1 = s[:, 14], s[:, 15]
        mid = np.linspace(t0.min(), t1.max(), 41)[1:-1]
        conc = [int(((t0 <= m) & (t1 >= m)).sum()) for m in mid]
        print(f"   wall clock: kernel span {(t1.max() - t0.min()) / 100:.1f} us, workgroup lifetime median {np.median(t1 - t0) / 100:.1f} us, "
              f"entry skew {(t0.max() - t0.min()) / 100:.1f} us, resident workgroups over time (39 samples): min {min(conc)} median {int(np.median(conc))} max {max(conc)}")
    for slot, name in (extra or {}).items():
        print(f"   wave 0, summed over the chunks: {name:36s} median {np.median(s[:, slot]):9.0f}  max {s[:, slot].max():9d}")
    last = np.max(s[:, 1:len(names)], axis=1)
    print(f"   workgroup lifetime: median {np.median(last - s[:, 0]):.0f} max {(last - s[:, 0]).max()} cycles")


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "roipool_fwd"
    st = torch.cuda.current_stream().cuda_stream
    if what == "roipool_fwd":
        R, C, H, W, K = 300, 1024, 38, 63, 7
        fm, out = torch.rand(C, H, W, device=dev), torch.empty(R, C, K, K, device=dev)
        rois = torch.from_numpy(bench_ops.random_rois(R, 0)).to(dev)
        run("d2t_lab_pool_fwd_stamps",
            lambda: bench_ops._check(lib.d2t_roipool_fwd_f32(fm.data_ptr(), rois.data_ptr(), out.data_ptr(), R, C, H, W, K, 0, 0, 0, st)),
            ["entry", "planes in LDS", "prefix2d done", "geometry 1 done", "look-ups 1 done", "geometry 2 done", "look-ups 2 done", "end"])
    if what == "roipool_direct":                                     # round 6: d2t_roipool_fwd_direct.hip
        R, C, H, W, K = [int(x) for x in sys.argv[2:6]] + [7] if len(sys.argv) > 5 else (300, 1024, 38, 63, 7)
        fm, out = torch.rand(C, H, W, device=dev), torch.empty(R, C, K, K, device=dev)
        rois = torch.from_numpy(bench_ops.random_rois(R, 0)).to(dev)
        call = lambda: bench_ops._check(lib.d2t_roipool_fwd_f32(fm.data_ptr(), rois.data_ptr(), out.data_ptr(), R, C, H, W, K, 0, 0, 0, st))
        dbg = lib.d2t_lab_roipool_direct_stamps_dbg
        dbg.restype, dbg.argtypes = ctypes.c_int, [ctypes.c_int]
        for bits, name in ((0, "product"), (1, "no stores"), (2, "no LDS reads in the walk"), (3, "neither")):
            assert dbg(bits) == 0
            print(f"--- ablation {bits}: {name}: op {bench_ops.timed(lambda i: call(), 20, 1):.1f} us (one buffer set: cache-warm)")
            run("d2t_lab_roipool_direct_stamps", call, ["entry", "planes + geometry in LDS", "bins walked, stored"])
        assert dbg(0) == 0
    if what == "roipool_bwd":
        R, C, H, W, K = 300, 1024, 38, 63, 7
        go, gin = torch.rand(R, C, K, K, device=dev), torch.empty(C, H, W, device=dev)
        rois = torch.from_numpy(bench_ops.random_rois(R, 0)).to(dev)
        nb = lib.d2t_roipool_bwd_workspace_bytes(R, C, H, W, K, 4)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        call = lambda: bench_ops._check(lib.d2t_roipool_bwd_f32(go.data_ptr(), rois.data_ptr(), gin.data_ptr(), R, C, H, W, K, ws.data_ptr(), nb, 0, st))
        dbg = lib.d2t_lab_pool_bwd_stamps_dbg
        dbg.restype, dbg.argtypes = ctypes.c_int, [ctypes.c_int]
        for bits, name in ((0, "product"), (1, "no gradOut loads"), (2, "no MFMA"), (4, "no list reads in the k-steps"), (3, "no loads, no MFMA"), (7, "none of the three")):
            assert dbg(bits) == 0
            print(f"--- ablation {bits}: {name}: op {bench_ops.timed(lambda i: call(), 20, 1):.1f} us")
            if bits in (0, 1, 7):
                run("d2t_lab_pool_bwd_stamps", call, ["entry", "k-steps done", "reduced + stored"])
        assert dbg(0) == 0
    if what.startswith("psroipool_bwd"):
        R, nT = (int(x) for x in (sys.argv[2], sys.argv[3])) if len(sys.argv) > 3 else (3000, 31)
        H, W, K = 38, 75, 7
        go, gin = torch.rand(R, nT, K, K, device=dev), torch.empty(nT * K * K, H, W, device=dev)
        rois = torch.from_numpy(bench_ops.random_rois(R, 1)).to(dev)
        nb = lib.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, K, 4)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        print(f"psroipool backward R={R} nT={nT} {H}x{W}")
        run("d2t_lab_pool_bwd_stamps",
            lambda: bench_ops._check(lib.d2t_psroipool_bwd_f32(go.data_ptr(), rois.data_ptr(), gin.data_ptr(), R, nT, H, W, K, ws.data_ptr(), nb, 0, st)),
            ["entry", "hit list done", "chunk 0 staged", "chunks done", "end"],
            extra={5: "load_chunk issue", 6: "k-steps (LDS reads + MFMA)", 7: "store_chunk (waits for the loads)", 8: "barrier", 9: "chunks"})


if __name__ == "__main__":
    main()
