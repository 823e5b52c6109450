import sys, json
sys.path.insert(0, "."); sys.path.insert(0, "detect-to-track_amd")
import bench_ops
for e in bench_ops.measure_tracker("cuda:0", 0, 40):
    print(e["op"], e.get("direction") or e.get("variant") or "", e.get("us"), flush=True)
