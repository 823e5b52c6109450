#!/usr/bin/env python3
"""CPU stand-in for bench.py's device layer -- TEST HARNESS, launched by tests/test_sharding_gloo.py as

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port P tests/bench_stub_main.py --gpus 2 --steps K --warmup W

(or bare, `python tests/bench_stub_main.py --gpus 2 ...`: bench.main() then launches the two ranks itself).
It runs bench.main() / bench.run() -- the real rank skeleton: WORLD_SIZE check, process-group init, setup, warmup,
barrier, K timed steps, barrier, MAX over ranks, one JSON line on rank 0, teardown -- with the two
op calls replaced by a sleep whose length depends on the rank, and gloo instead of RCCL.  The HIP
library is not loaded; nothing here is reachable from bench.py itself.
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


class CpuStub:
    backend = "gloo"
    FWD_S, BWD_S = 0.002, 0.003

    def __init__(self):
        self.rank = bench.rank_env()[0]
        self.calls = {"fwd": 0, "bwd": 0}

    def init_process_group(self):
        import torch.distributed as dist
        dist.init_process_group(self.backend)

    def setup(self, cfg, n_sets, seed):
        assert seed == self.rank
        self.n_sets = n_sets

    def fwd(self, i):
        assert 0 <= i < self.n_sets
        self.calls["fwd"] += 1
        time.sleep(self.FWD_S * (1 + self.rank))               # rank 1 is the slow shard

    def bwd(self, i):
        assert 0 <= i < self.n_sets
        self.calls["bwd"] += 1
        time.sleep(self.BWD_S * (1 + self.rank))

    def capture(self, order):
        """bench.run()'s graph branch: on the GPU one HIP graph of the K steps; here the list of steps."""
        assert all(0 <= z < self.n_sets for z in order)
        return list(order)

    def replay(self, graph):
        for z in graph:
            self.fwd(z)
            self.bwd(z)

    def synchronize(self):
        pass

    def new_event(self):
        return [0.0]

    def record(self, ev):
        ev[0] = time.perf_counter()

    def elapsed_ms(self, e0, e1):
        return (e1[0] - e0[0]) * 1e3

    def reduce_device(self):
        return None


if __name__ == "__main__":
    # The same entry as `python3 bench.py ...`: under torch.distributed.run this process is a rank; launched bare with
    # --gpus N > 1 bench.main() starts the N ranks of THIS script itself (bench.launch_ranks) and exits with their code.
    made = []

    def factory():
        made.append(CpuStub())
        return made[0]

    import os
    argv = sys.argv[1:] + ([] if os.environ.get("D2T_STUB_CPU_BASELINE") == "1" or "--no-cpu-baseline" in sys.argv[1:] else ["--no-cpu-baseline"])
    info = bench.main(argv, device_factory=factory, script=Path(__file__).resolve())
    a, stub = bench.parse_args(argv), made[0]
    settle = info["settle_steps"]                                # untimed clock-settling steps (bounded by wall time, in batches)
    assert settle % bench.SETTLE_BATCH == 0 and (settle > 0) == (a.settle_ms > 0)
    n = stub.n_sets + a.warmup + settle + a.steps * (4 if a.graph else 2)   # + the timed steps and the event pass (+ 2 graph replays of them)
    assert stub.calls == {"fwd": n, "bwd": n}, stub.calls
    assert info["ranks_seen"] == a.gpus
    # rank -> device: what HipDevice would pick with one GPU per rank (RCCL) and when all ranks share one GPU (gloo rehearsal)
    _, world, lr = bench.rank_env()
    print(f"stub rank {stub.rank} done local_rank {lr} nccl_device {bench.device_index(lr, 'nccl', world)} "
          f"shared_device {bench.device_index(lr, 'gloo', 1)}", file=sys.stderr)
