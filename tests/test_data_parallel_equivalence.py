"""GPU, ONE MI355X: data-parallel EQUIVALENCE of the real training step (BASELINE config 5).

The reference sums the losses of a minibatch's pairs, divides by the pair count and runs one backward
(/root/reference/detect_to_track/trainer.py:258-276, utils.py:64-75).  Sharding the pairs over ranks and MEAN-reducing the
gradients must give the same gradient.  Two ranks (fresh child processes under ``torch.distributed.run``, gloo, sharing the one
GPU) run ``DataParallelTrainer.train_step`` on disjoint shards of 2P synthetic pairs from the SAME initial weights; one process
runs all 2P pairs; every trainable parameter's averaged gradient must equal the single-process gradient within 1e-5 (relative to
the parameter's largest gradient entry), and so must the mean of the five loss terms.  tests/dp_equivalence_main.py is the child.
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
CHILD = str(ROOT / "tests" / "dp_equivalence_main.py")


def _env():
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(p):
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_two_ranks_mean_gradient_equals_single_process_gradient(tmp_path):
    ref = str(tmp_path / "single.pt")
    P, world = 2, 2
    s = _json_line(subprocess.run([sys.executable, CHILD, "--mode", "single", "--pairs-per-rank", str(P), "--world", str(world), "--out", ref],
                                  env=_env(), capture_output=True, text=True, timeout=500, cwd=str(ROOT)))
    assert s["pairs"] == world * P and s["params"] >= 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), CHILD, "--mode", "ranks", "--pairs-per-rank", str(P), "--world", str(world), "--ref", ref,
           "--rtol", "1e-5"]
    d = _json_line(subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=str(ROOT)))
    assert d["params"] == s["params"] and d["buckets"] >= 3
    assert d["worst_rel_err"] <= 1e-5, d                    # every trainable parameter, on both ranks
    assert d["loss_rel_err"] <= 1e-5, d
    assert d["all_ranks_ok"] is True, d
