"""GPU, ONE MI355X: the RCCL path of bench.py and bench_model.py at world size 1.

No multi-GPU box is available to the builder, so the N > 1 path can only be covered with gloo on the CPU
(tests/test_sharding_gloo.py, tests/test_data_parallel_gloo.py).  What CAN run on one GPU is everything except the second rank:
`--force-dist` makes both benchmarks create the "nccl" (= RCCL) process group at world size 1 under `torch.distributed.run
--nproc-per-node 1` and run their barriers, MAX / SUM reductions and -- in bench_model.py -- the bucketed asynchronous gradient
all-reduces launched from autograd hooks as REAL RCCL collectives on GPU tensors and torch's streams.  It proves the plumbing
(HipDevice.init_process_group, HSA_ENABLE_IPC_MODE_LEGACY, stream ordering of the hooks, teardown); it measures nothing about xGMI.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun_one(script, *argv, timeout=500, nproc=1):
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / script), *argv]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_collectives_over_rccl_world_size_1():
    d = _torchrun_one("bench.py", "--gpus", "1", "--force-dist", "1", "--steps", "5", "--warmup", "2", "--settle-ms", "20",
                      "--no-cpu-baseline", "--graph", "0", "--extras", "0", "--ops", "0")
    assert d["process_group"] == "nccl" and d["n_gpus"] == 1 and d["ranks_seen"] == 1
    assert d["value"] > 1.0 and d["kernels"][1]["us"] > d["kernels"][0]["us"] > 10.0      # the real kernels ran between real barriers


@pytest.mark.timeout(900)
def test_gradient_buckets_over_rccl_world_size_1():
    d = _torchrun_one("bench_model.py", "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--height", "320", "--width", "480",
                      "--rois", "64", timeout=800)
    assert d["process_group"] == "nccl" and d["gradient_buckets"] is True and d["finite"] is True
    assert d["custom_ops_calls_per_step"]["pointwise_correlation_levels_forward"] == 1


# ---- two ranks on the ONE GPU, collectives over gloo: the N > 1 code path with the real kernels in both processes.  Not a
# scaling figure (the ranks share the chip) and not RCCL -- but every branch bench.py / bench_model.py take at world size 2 runs:
# per-rank shards and seeds, barriers around the timed region, MAX / SUM over two ranks, rank 0 alone printing, and in
# bench_model.py the bucketed gradient all-reduce averaging two ranks' real gradients.

@pytest.mark.timeout(600)
def test_bench_two_ranks_share_the_gpu_over_gloo():
    d = _torchrun_one("bench.py", "--gpus", "2", "--backend", "gloo", "--steps", "5", "--warmup", "2", "--settle-ms", "20",
                      "--no-cpu-baseline", "--graph", "0", "--extras", "0", "--ops", "0", nproc=2)
    assert d["process_group"] == "gloo" and d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["gpus_visible"] == 1
    assert d["scaling"] == "weak" and d["value"] > 1.0                        # both shards' voxels over the slower rank's time
    assert d["kernels"][1]["us"] > d["kernels"][0]["us"] > 10.0


@pytest.mark.timeout(900)
def test_data_parallel_step_two_ranks_share_the_gpu_over_gloo():
    d = _torchrun_one("bench_model.py", "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--height", "320", "--width", "480",
                      "--rois", "64", timeout=800, nproc=2)
    assert d["process_group"] == "gloo" and d["n_gpus"] == 2 and d["gradient_buckets"] is True and d["finite"] is True
    assert d["custom_ops_calls_per_step"]["pointwise_correlation_levels_forward"] == 1
