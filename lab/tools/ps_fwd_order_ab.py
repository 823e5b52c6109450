import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "detect-to-track_amd")
import bench_ops
st = torch.cuda.current_stream().cuda_stream
for (R, nT, H, W) in ((300, 21, 38, 63), (300, 31, 38, 75), (1000, 31, 38, 75), (700, 16, 38, 63)):
    row = []
    for name, v in (("in the kernel", "0"), ("pre-pass", "1"), ("in the kernel", "0"), ("pre-pass", "1")):
        os.environ["D2T_PS_FWD_PREPASS"] = v
        e = bench_ops.measure_psroipool("cuda:0", R, nT, H, W, 0, 40, st)
        row.append(f"{name} {e[0]['us']:.1f}")
    print(f"R={R} nT={nT} {H}x{W} fwd [us]: " + " | ".join(row), flush=True)
