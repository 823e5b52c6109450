"""CPU, world_size 2, gloo: the multi-GPU path of bench.py.

The hot path shards by frame pair with NO exchange step: a rank's outputs
depend only on its own shard.  The only cross-rank operations in bench.py are the timing barrier
and the MAX-over-ranks reduction; both are exercised here over gloo, together with the
shard-independence property itself (checked with the CPU oracle: test infrastructure).
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "detect-to-track_amd")]
    import bench
    import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert bench.rank_env() == (rank, world, rank)
        # (1) MAX over ranks of the elapsed time: rank r "takes" 1 + r seconds
        elapsed = bench.max_over_ranks(1.0 + rank, world)
        assert elapsed == float(world)
        # (2) whole-job value counts every rank's shard once
        assert bench.whole_job_value(100, world, 10, elapsed) == world * 100 * 10 / elapsed
        # (3) shard independence: each rank runs the oracle on ITS pairs only; gathered, the result
        #     equals the full-batch computation bit for bit -> no data-path collective is needed
        rng = np.random.default_rng(7)
        B, C, H, W, d = 4, 6, 9, 21, 8
        fm0, fm1 = rng.random((B, C, H, W), dtype=np.float32), rng.random((B, C, H, W), dtype=np.float32)
        lo, hi = rank * B // world, (rank + 1) * B // world
        mine = torch.from_numpy(O.corr_fwd(fm0[lo:hi], fm1[lo:hi], d, 1))
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)                       # test-side gather only, to compare
        full = O.corr_fwd(fm0, fm1, d, 1)
        assert np.array_equal(torch.cat(parts).numpy(), full)
        dist.barrier()
        q.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(world)) == [0, 1]


def test_corr_counts_match_baseline_md():
    sys.path.insert(0, str(ROOT))
    import bench
    c = bench.corr_counts(**bench.WORKLOADS["corr_B8_C256_38x63_d8"])
    assert c["vox"] == 5534928 and c["fwd_bytes"] == 61363008 and c["bwd_bytes"] == 100586304
    assert c["fwd_flops"] == 2103443456 and c["bwd_flops"] == 2 * 2103443456


@pytest.mark.timeout(300)
def test_bench_rank_skeleton_under_torchrun():
    """bench.run() end to end under `torch.distributed.run --nproc-per-node 2` exactly as the driver
    launches bench.py for N > 1 (rendezvous on 127.0.0.1), with the device layer replaced BY THIS
    TEST's harness (tests/bench_stub_main.py: gloo, op calls = rank-dependent sleeps)."""
    K, W = 4, 2
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "tests" / "bench_stub_main.py"), "--gpus", "2", "--steps", str(K), "--warmup", str(W)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                        # rank 0 prints ONE line, rank 1 none
    d = json.loads(lines[0])
    sys.path.insert(0, str(ROOT))
    import bench
    assert d["metric"] == bench.METRIC and d["n_gpus"] == 2 and d["steps"] == K and d["warmup"] == W
    assert d["scaling"] == "weak" and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "shard2"
    assert d["ranks_seen"] == 2 and d["settle_ms"] >= 100.0
    # the slow rank (rank 1 sleeps 2x) sets the time: MAX over ranks, not rank 0's own clock
    assert d["ms_per_step"] >= 2 * (2 + 3) * 0.95, d["ms_per_step"]
    vox = bench.corr_counts(**bench.WORKLOADS["corr_B8_C256_38x63_d8"])["vox"]
    assert abs(d["value"] - 2 * vox / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]
    # value: K steps with nothing else on the stream; kernels[]: the event intervals of a second pass, which tile THAT pass;
    # the graph replay is an extra, not the metric
    assert sum(k["us"] for k in d["kernels"]) <= d["event_pass"]["ms_per_step"] * 1e3 * 1.001
    assert d["event_pass"]["records_per_step"] == 2
    assert d["timing"]["value"].startswith("wall time of K eager steps") and d["graph_replay"]["ms_per_step"] >= 2 * (2 + 3) * 0.95
    assert "cpu_baseline" not in d                          # the stub passes --no-cpu-baseline (see the N = 2 case below)
    # round-4 fields: the third roof per kernel (profiles/ta_roof.json), the op table (device layers without one: null), the backend
    for k in d["kernels"]:
        ta = k["roofline_ta"]
        assert ta is None or (ta["lines"] > 0 and ta["peak"] > 0 and abs(ta["frac"] - ta["achieved"] / ta["peak"]) < 1e-9)
    assert d["ops"] is None and d["process_group"] == "gloo"
    # round-6 fields: where `traffic` came from (never this run: PMC needs rocprofv3 around the process), the launch duration with the
    # event record taken out beside the raw interval, and the forward -- the kernel the metric string names -- against both roofs
    r, rf = d["roofline"], d["roofline_fwd"]
    assert r["traffic_source"] is None or (r["traffic_source"]["measured_in_this_run"] is False and r["traffic_source"]["file"] == "profiles/traffic.json")
    assert r["launch_us"] <= r["launch_us_event_interval"] and r["frac"] >= r["frac_event_interval"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert rf["kernel"] == "corr_fwd" and rf["pct_hbm"] > 0 and rf["pct_f32"] > 0 and rf["target_pct_hbm"] == 70.0
    assert abs(rf["pct_hbm"] - 100 * rf["hbm"]["achieved"] / rf["hbm"]["peak"]) < 1e-6
    assert p.stderr.count("stub rank") == 2


@pytest.mark.timeout(300)
def test_bench_bare_form_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` the way the driver launches N = 1 -- no torch.distributed.run in front, no
    WORLD_SIZE in the environment: the GPU-free parent (bench.main -> bench.launch_ranks) starts the two ranks as a child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...` and hands back their exit
    code; rank 0's line is the only JSON line.  Device layer = this test's stub, as above."""
    K, W = 3, 1
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "tests" / "bench_stub_main.py"), "--gpus", "2", "--steps", str(K), "--warmup", str(W),
           "--settle-ms", "30"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == K and d["config"]["global_batch"] == 16
    assert d["settle_steps"] >= 25 and d["settle_ms"] >= 30.0
    assert p.stderr.count("stub rank") == 2                 # both ranks ran to the end


@pytest.mark.timeout(600)
def test_bench_eight_ranks_one_line_and_device_map():
    """World size 8 (BASELINE config 5's rank count) with the stub device over gloo, bare form: eight ranks take part
    (ranks_seen), rank 0 prints the one JSON line, LOCAL_RANK r is device r under RCCL's one-GPU-per-rank rule and device 0
    when eight ranks rehearse on one GPU; a rank without a GPU of its own is refused, not wrapped."""
    K, W = 2, 1
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "tests" / "bench_stub_main.py"), "--gpus", "8", "--steps", str(K), "--warmup", str(W),
           "--settle-ms", "10", "--graph", "0"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "shard8"
    assert d["scaling"] == "weak" and d["ms_per_step"] >= 8 * (2 + 3) * 0.95          # the slowest rank (rank 7: 8x sleeps) sets the time
    done = [ln for ln in p.stderr.splitlines() if ln.startswith("stub rank")]
    assert len(done) == 8
    seen = sorted((int(w[2]), int(w[5]), int(w[7]), int(w[9])) for w in (ln.split() for ln in done))
    assert seen == [(r, r, r, 0) for r in range(8)], seen                              # (rank, LOCAL_RANK, RCCL device, shared device)
    sys.path.insert(0, str(ROOT))
    import bench
    with pytest.raises(SystemExit):
        bench.device_index(1, "nccl", 1)                                               # two RCCL ranks on one GPU: refused


@pytest.mark.timeout(300)
def test_bench_two_ranks_carry_a_cpu_baseline():
    """N > 1: rank 0's line carries `cpu_baseline` too (north_star: the CPU figure beside the 2 / 4 / 8-GPU numbers) -- a shorter sample on
    rank 0's share of the host threads, taken after everything timed."""
    env = dict(os.environ, OMP_NUM_THREADS="1", D2T_STUB_CPU_BASELINE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "tests" / "bench_stub_main.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--settle-ms", "10",
           "--graph", "0", "--cpu-baseline-multi-s", "1.0"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    cb = d["cpu_baseline"]
    assert d["n_gpus"] == 2 and cb["kind"] == "port" and cb["unit"] == "Gvox/s" and cb["value"] > 0 and cb["cores"] >= 1
    assert "rank 0 of 2" in cb["sample"] and cb["single_thread"]["cores"] == 1


@pytest.mark.timeout(300)
def test_bench_force_dist_world_size_1():
    """--force-dist 1 under `torch.distributed.run --nproc-per-node 1`: the process group is created and every barrier / MAX /
    SUM is a real collective although there is one rank (the GPU counterpart is tests/test_rccl_single_gpu.py)."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "bench_stub_main.py"), "--gpus", "1", "--force-dist", "1",
           "--steps", "2", "--warmup", "1", "--settle-ms", "10"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["process_group"] == "gloo" and d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["config"]["global_batch"] == 8


def test_bench_parent_never_touches_the_gpu():
    """The bare-form parent must start the ranks BEFORE anything imports torch or initialises HIP (a process that has
    touched the GPU must not spawn / be replaced by ranks on this pool): bench.py imports torch only inside functions
    the parent does not call, and a failing child's exit code is handed back."""
    src = (ROOT / "bench.py").read_text()
    head = src.split("def corr_counts")[0]
    assert "import torch" not in head                       # module level: no torch
    sys.path.insert(0, str(ROOT))
    import bench
    code = bench.launch_ranks(ROOT / "tests" / "no_such_script.py", [], 2)
    assert code != 0


def test_bench_rank_refuses_mismatched_world(monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench
    monkeypatch.setenv("WORLD_SIZE", "1")
    a = bench.parse_args(["--gpus", "2"])
    with pytest.raises(SystemExit):
        bench.run(a, object())                              # a rank whose WORLD_SIZE disagrees with --gpus: refused before any device call
