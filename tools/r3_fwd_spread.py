"""dev helper: per-launch forward / backward event intervals of bench.py's step loop, by buffer set."""
import sys
sys.path.insert(0, ".")
import bench, types
args = bench.parse_args(["--steps", "60", "--warmup", "10", "--no-cpu-baseline"])
dev = bench.HipDevice(0, 0)
cfg = bench.WORKLOADS[args.workload]
dev.setup(cfg, 6, seed=0)
for i in range(60):
    dev.fwd(i % 6); dev.bwd(i % 6)
dev.synchronize()
K = 60
ev = [dev.new_event() for _ in range(2 * K + 1)]
dev.record(ev[0])
for i in range(K):
    dev.fwd(i % 6); dev.record(ev[2 * i + 1]); dev.bwd(i % 6); dev.record(ev[2 * i + 2])
dev.synchronize()
f = [dev.elapsed_ms(ev[2 * i], ev[2 * i + 1]) * 1e3 for i in range(K)]
b = [dev.elapsed_ms(ev[2 * i + 1], ev[2 * i + 2]) * 1e3 for i in range(K)]
for s in range(6):
    fs = [f[i] for i in range(K) if i % 6 == s]; bs = [b[i] for i in range(K) if i % 6 == s]
    z = dev.sets[s]
    print(f"set {s}: fwd mean {sum(fs)/len(fs):5.1f} min {min(fs):5.1f} max {max(fs):5.1f} | bwd mean {sum(bs)/len(bs):5.1f} min {min(bs):5.1f} max {max(bs):5.1f} | "
          f"fm0 @{z['fm0'].data_ptr() % (1<<21):#x} fm1 @{z['fm1'].data_ptr() % (1<<21):#x} out @{z['out'].data_ptr() % (1<<21):#x}")
print("fwd sequence:", " ".join(f"{x:.0f}" for x in f[:36]))
