import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "detect-to-track_amd")
import bench_ops
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    e = bench_ops.measure_roipool("cuda:0", 300, 1024, 38, 63, 0, 50, st)
    print("R300 C1024 38x63 fwd", round(e[0]["us"], 1), flush=True)
e = bench_ops.measure_roipool("cuda:0", 300, 1891, 38, 75, 0, 50, st)
print("R300 C1891 38x75 fwd", round(e[0]["us"], 1), flush=True)
