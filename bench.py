#!/usr/bin/env python3
"""bench.py -- headline benchmark of the detect-to-track custom-op hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: PointwiseCorrelation
forward + backward at BASELINE.json's metric shape (B=8, C=256, 38x63, d_max=8, stride 1, f32),
called through the C ABI of libd2t_ops.so exactly as the autograd Function calls it (same entry
points, caller-allocated outputs, torch's current stream).  Inputs are resident in HBM before the
timed region.  `--sets` independent buffer sets are rotated (default: > 512 MiB footprint) so the
256 MiB Infinity Cache does not stand in for HBM; every set is touched once during setup (page
tables, clocks) before the W warmup steps.

Multi-GPU: the path shards by frame-pair with no exchange step, so every rank runs the same
per-GPU workload on its own shard (weak scaling) with NO data-path collective; ranks only meet at
the timing barriers and the MAX-over-ranks reduction of the elapsed time.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).

The rank skeleton (`run`) takes its device layer as an object: `HipDevice` (below) is the real
one; tests/bench_stub_main.py passes a CPU stand-in so that the launch / init / barrier /
MAX-over-ranks / print-on-rank-0 path runs under torch.distributed.run with gloo on a box
without GPUs (tests/test_sharding_gloo.py).  Nothing in this file selects a stub by itself.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

# The host driver of this pool only supports dmabuf IPC: without this RCCL (and CUDA-tensor sharing across processes) fails
# with "hipIpcGetMemHandle: invalid argument".  Must be in the environment before HIP initialises; harmless elsewhere.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP32_MATRIX_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_* dense f32 = f32 vector peak
METRIC = "PointwiseCorrelation fwd+bwd Gvox/s (and % HBM roofline) at B=8 C=256 38×63 d=8"

WORKLOADS = {
    # BASELINE.json metric shape ("north star")
    "corr_B8_C256_38x63_d8": dict(B=8, C=256, H=38, W=63, d=8, s=1),
    # BASELINE.json configs[1]
    "corr_B1_C256_38x63_d8": dict(B=1, C=256, H=38, W=63, d=8, s=1),
    # shapes the reference model really produces (correlation_tracker.py:57-70)
    "corr_B1_C512_38x75_d8": dict(B=1, C=512, H=38, W=75, d=8, s=1),
    "corr_B1_C1024_38x75_d8": dict(B=1, C=1024, H=38, W=75, d=8, s=1),
    "corr_B1_C2048_38x75_d8": dict(B=1, C=2048, H=38, W=75, d=8, s=1),
}


def corr_counts(B, C, H, W, d, s):
    """Algorithmic bytes / flops per launch (DESIGN.md "Measurement")."""
    cw = 2 * d + 1
    vox = B * H * W * cw * cw
    cells = 0                                     # window cells the reference actually computes
    for i in range(H):
        ni = len(range(max(0, i - d), min(i + d, H), s))
        for j in range(W):
            cells += ni * len(range(max(0, j - d), min(j + d, W), s))
    in_b = B * C * H * W * 4
    out_b = vox * 4
    return dict(vox=vox, fwd_bytes=2 * in_b + out_b, bwd_bytes=out_b + 2 * in_b + 2 * in_b,
                fwd_flops=2 * B * cells * C, bwd_flops=4 * B * cells * C)


def host_threads():
    n = len(os.sched_getaffinity(0))
    try:                                          # cgroup v2 CPU quota, if any
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(cfg, counts, budget_s=12.0, threads=None):
    """Time the CPU oracle (a port: the reference has no CPU path, common/cpp_common.hpp:1) on
    the same workload: all host threads (the headline `value`) and one thread on a
    quarter-batch sample.  Checker code, used here ONLY as a baseline."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import numpy as np
    threads = threads or host_threads()
    import oracle as O
    O.set_threads(threads)
    rng = np.random.default_rng(0)
    shp = (cfg["B"], cfg["C"], cfg["H"], cfg["W"])
    cw = 2 * cfg["d"] + 1
    fm0, fm1 = rng.random(shp, dtype=np.float32), rng.random(shp, dtype=np.float32)
    g = rng.random((cfg["B"], cfg["H"], cfg["W"], cw, cw), dtype=np.float32)
    O.corr_fwd(fm0[:1], fm1[:1], cfg["d"], cfg["s"])                    # load + warm
    reps, t0 = 0, time.perf_counter()
    while True:
        O.corr_fwd(fm0, fm1, cfg["d"], cfg["s"])
        O.corr_bwd(g, fm0, fm1, cfg["d"], cfg["s"])
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 50:
            break
    # one thread: a bounded sample (the first nb pairs) of the same batch, one fwd+bwd
    nb = max(1, cfg["B"] // 4)
    O.set_threads(1)
    t1 = time.perf_counter()
    O.corr_fwd(fm0[:nb], fm1[:nb], cfg["d"], cfg["s"])
    O.corr_bwd(g[:nb], fm0[:nb], fm1[:nb], cfg["d"], cfg["s"])
    el1 = time.perf_counter() - t1
    O.set_threads(threads)
    return dict(value=counts["vox"] * reps / el / 1e9, unit="Gvox/s", cores=threads, kind="port",
                sample=f"{reps} full steps (fwd+bwd, B={cfg['B']}) of the same workload in {el:.1f} s, "
                       f"oracle/libd2t_oracle.so with OpenMP over {threads} threads",
                single_thread=dict(value=counts["vox"] * nb / cfg["B"] / el1 / 1e9, unit="Gvox/s", cores=1,
                                   sample=f"1 step (fwd+bwd) on the first {nb} of {cfg['B']} pairs in {el1:.1f} s"))


def rank_env():
    """(rank, world, local_rank) as torch.distributed.run exports them; (0, 1, 0) when launched bare."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def max_over_ranks(seconds, world, device=None):
    """The timed region ends when the slowest rank is done: MAX over ranks of the elapsed time."""
    if world == 1:
        return seconds
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


SETTLE_MS = 100.0     # untimed steps in front of the timed region until this much wall time has passed: the clocks of a fresh
SETTLE_BATCH = 25     # box need ~50 ms of this load to settle (checked every SETTLE_BATCH steps; K = 20 and K = 200 then measure
SETTLE_MAX_STEPS = 4000   # the same clocks)


def sum_over_ranks(x, world, device=None):
    """SUM over ranks (ranks_seen = an all-reduce of ones: how many ranks really took part)."""
    if world == 1:
        return x
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def launch_ranks(script, argv, n):
    """`python3 bench.py --gpus N` launched bare (N > 1, no WORLD_SIZE in the environment): start the N ranks the way the
    driver does -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per
    GPU -- as a CHILD process, and hand its exit code back.  This parent never imports torch and never touches HIP (a
    process that has initialised the GPU must not be replaced or forked into ranks on this pool); the ranks inherit stdout,
    so rank 0's single JSON line is the only line printed."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # --standalone: torchrun's own c10d rendezvous on a port it binds itself (no bind-close-reuse race between this parent
    # and the ranks); --local-addr 127.0.0.1 because the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), str(script), *argv]
    return subprocess.call(cmd, env=env)


def whole_job_value(units_per_rank_step, world, steps, elapsed_s):
    """Aggregate throughput: every rank processes its own shard (weak scaling, no exchange)."""
    return world * units_per_rank_step * steps / elapsed_s


def device_index(local_rank, backend, device_count):
    """Which GPU a rank uses.  RCCL ("nccl"): one GPU per rank, LOCAL_RANK is the device (a rank without a GPU of its own is an
    error, not a wrap-around: two ranks on one device would deadlock RCCL).  Rehearsal backends (gloo): ranks may share GPUs."""
    if backend == "nccl":
        if not 0 <= local_rank < device_count:
            raise SystemExit(f"rank with LOCAL_RANK {local_rank} has no GPU of its own ({device_count} visible): "
                             "launch one rank per GPU, or rehearse with --backend gloo")
        return local_rank
    return local_rank % max(1, device_count)


class HipDevice:
    """The device layer of the benchmark: buffers in HBM, the two C-ABI calls, HIP events on the
    launch stream, RCCL process group.  (`torch.distributed` backend "nccl" is RCCL on ROCm.)"""
    backend = "nccl"

    def __init__(self, local_rank, impl, backend="nccl"):
        import torch
        from detect_to_track.models import _native       # raises ImportError if the HIP library is missing
        self.torch, self.native, self.lib, self.impl, self.backend = torch, _native, _native.lib, impl, backend
        local_rank = device_index(local_rank, backend, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)

    def gpus_visible(self):
        return self.torch.cuda.device_count()

    def is_lab_build(self):
        return bool(getattr(self.native, "IS_LAB_BUILD", False))

    def init_process_group(self):
        import torch.distributed as dist
        if self.backend == "nccl":
            dist.init_process_group("nccl", device_id=self.dev)
        else:
            dist.init_process_group(self.backend)

    def setup(self, cfg, n_sets, seed):
        torch = self.torch
        B, C, H, W, d, s = (cfg[k] for k in "BCHWds")
        cw = 2 * d + 1
        torch.manual_seed(seed)                                # per-rank shard of synthetic frame pairs
        self.cfg = cfg
        self.sets = [dict(fm0=torch.rand(B, C, H, W, device=self.dev), fm1=torch.rand(B, C, H, W, device=self.dev),
                          gout=torch.rand(B, H, W, cw, cw, device=self.dev),
                          out=torch.empty(B, H, W, cw, cw, device=self.dev),
                          g0=torch.empty(B, C, H, W, device=self.dev), g1=torch.empty(B, C, H, W, device=self.dev))
                     for _ in range(n_sets)]
        self.wsf_n = self.lib.d2t_corr_fwd_workspace_bytes(B, C, H, W, d, s, 4)
        self.wsb_n = self.lib.d2t_corr_bwd_workspace_bytes(B, C, H, W, d, s, 4)
        self.wsf = torch.empty(max(self.wsf_n, 1), dtype=torch.uint8, device=self.dev)
        self.wsb = torch.empty(max(self.wsb_n, 1), dtype=torch.uint8, device=self.dev)
        self.stream = torch.cuda.current_stream(self.dev)
        self.sh = self.stream.cuda_stream

    def _check(self, rc):
        if rc:
            raise RuntimeError(self.native.error_string(rc).decode())

    def fwd(self, i):
        z, c = self.sets[i], self.cfg
        self._check(self.lib.d2t_corr_fwd_f32(z["fm0"].data_ptr(), z["fm1"].data_ptr(), z["out"].data_ptr(),
                                              c["B"], c["C"], c["H"], c["W"], c["d"], c["s"],
                                              self.wsf.data_ptr(), self.wsf_n, self.impl, self.sh))

    def bwd(self, i):
        z, c = self.sets[i], self.cfg
        self._check(self.lib.d2t_corr_bwd_f32(z["gout"].data_ptr(), z["fm0"].data_ptr(), z["fm1"].data_ptr(),
                                              z["g0"].data_ptr(), z["g1"].data_ptr(),
                                              c["B"], c["C"], c["H"], c["W"], c["d"], c["s"],
                                              self.wsb.data_ptr(), self.wsb_n, self.impl, self.sh))

    def bwd_as(self, i, impl):
        """The backward of buffer set i with another implementation selector (extra measurements only)."""
        saved, self.impl = self.impl, impl
        try:
            self.bwd(i)
        finally:
            self.impl = saved

    def capture(self, order):
        """The steps `order` (buffer-set indices) as ONE HIP graph."""
        torch = self.torch
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            saved, self.sh = self.sh, torch.cuda.current_stream(self.dev).cuda_stream   # launch on the capturing stream
            try:
                for z in order:
                    self.fwd(z)
                    self.bwd(z)
            finally:
                self.sh = saved
        return graph

    def replay(self, graph):
        graph.replay()

    def snapshot(self, i):
        """Copies of the outputs of buffer set i (to compare a graph replay with eager launches)."""
        z = self.sets[i]
        return [z[k].clone() for k in ("out", "g0", "g1")]

    def same(self, a, b):
        return all(self.torch.equal(x, y) for x, y in zip(a, b))

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)

    def new_event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def record(self, ev):
        ev.record(self.stream)                                 # on the stream the kernels are launched on

    def elapsed_ms(self, e0, e1):
        return e0.elapsed_time(e1)

    def reduce_device(self):
        return self.dev

    def ops_extra(self):
        import bench_ops                                       # imports torch + the HIP library: never at module level here
        return bench_ops.measure(self.dev, impl=self.impl, iters=20)


def run(args, device):
    """The rank skeleton: init, setup, W warmup steps, EXACTLY K timed steps between barriers,
    MAX over ranks, one JSON line on rank 0."""
    import torch.distributed as dist

    rank, world, _ = rank_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: main() starts the ranks itself when launched bare")
    # --force-dist 1: take the process-group path at world size 1 too (under torch.distributed.run --nproc-per-node 1): the
    # RCCL group is created and every barrier / MAX / SUM below runs as a real collective on GPU tensors -- the only way to
    # exercise HipDevice.init_process_group and the collectives' stream semantics on a box with ONE GPU.
    dist_on = world > 1 or bool(getattr(args, "force_dist", 0))
    if dist_on:
        device.init_process_group()
    nred = world if world > 1 else (2 if dist_on else 1)               # > 1 makes the helpers below reduce
    ranks_seen = int(round(sum_over_ranks(1.0, nred, device.reduce_device())))

    cfg = WORKLOADS[args.workload]
    B, C, H, W = (cfg[k] for k in "BCHW")
    cnt = corr_counts(**cfg)
    set_bytes = 4 * (B * C * H * W * 4) + 2 * (cnt["vox"] * 4)            # fm0, fm1, g0, g1 + out, gout
    n_sets = args.sets or max(2, -(-(512 << 20) // set_bytes) + 1)
    device.setup(cfg, n_sets, seed=rank)

    def barrier():
        if dist_on:
            dist.barrier()
        device.synchronize()

    for i in range(n_sets):                                    # setup: touch every buffer set once
        device.fwd(i)
        device.bwd(i)
    for i in range(args.warmup):
        device.fwd(i % n_sets)
        device.bwd(i % n_sets)

    K = args.steps
    order = [(args.warmup + i) % n_sets for i in range(K)]
    # Untimed and bounded BY TIME, not by K: the box is fresh and the W warmup steps are few -- the chip ramps up under this
    # load for ~50 ms -- so the same steps run untimed until args.settle_ms of wall time have passed (the driver's K = 20
    # line of round 3 had min(3K, 400) = 60 settle steps = 7 ms and sat 8 % below the K = 200 line for that reason).
    settle_steps, t_settle = 0, time.perf_counter()
    while args.settle_ms > 0:
        for i in range(SETTLE_BATCH):
            device.fwd(order[(settle_steps + i) % K])
            device.bwd(order[(settle_steps + i) % K])
        settle_steps += SETTLE_BATCH
        device.synchronize()
        if (time.perf_counter() - t_settle) * 1e3 >= args.settle_ms or settle_steps >= SETTLE_MAX_STEPS:
            break
    settle_ms = (time.perf_counter() - t_settle) * 1e3
    # The timed region (the metric): EXACTLY K steps launched on the stream between two barriers (+ device synchronize),
    # nothing else on the stream.  value = vox of the K steps / wall time between the barriers.
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        device.fwd(order[i])
        device.bwd(order[i])
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, nred, device.reduce_device())
    # The per-kernel figures: the SAME K steps once more, with a HIP event in front of every kernel and one behind the
    # last (2 K + 1 records; a step's closing event is the next one's opening event).  kernels[] / roofline = the event
    # intervals; they add up to the wall time of THIS pass (asserted below), which is longer than the metric's: an event
    # record between two kernels costs ~3.7 us of an idle GPU on this stack (event_record_overhead_us = the difference
    # of the two passes per record), and an interval contains one.  Round 3a timed the metric in the event pass itself
    # (one pass for everything): that understated the throughput of K back-to-back steps by 6 %.
    ev = [device.new_event() for _ in range(2 * K + 1)]
    barrier()
    t0 = time.perf_counter()
    device.record(ev[0])
    for i in range(K):
        z = order[i]
        device.fwd(z)
        device.record(ev[2 * i + 1])
        device.bwd(z)
        device.record(ev[2 * i + 2])
    barrier()
    elapsed_ev = time.perf_counter() - t0
    us_fwd = [device.elapsed_ms(ev[2 * i], ev[2 * i + 1]) * 1e3 for i in range(K)]
    us_bwd = [device.elapsed_ms(ev[2 * i + 1], ev[2 * i + 2]) * 1e3 for i in range(K)]
    elapsed_ev = max_over_ranks(elapsed_ev, nred, device.reduce_device())

    # Extra, NOT the metric (--graph 1): the same K steps replayed as one HIP graph -- what a training loop that captures
    # its steps gets (no event records, no host work between the kernels).  The C-ABI calls are asynchronous, allocate
    # nothing and keep no state, so they capture; the replay is checked bit for bit against eager launches.  (Per-kernel
    # events cannot be recorded inside a captured graph on this stack: torch refuses external events on ROCm and
    # hipEventRecordWithFlags(..., hipEventRecordExternal) returns hipErrorInvalidValue, lab/tools/r3_extev.py.)
    graph_replay = None
    if args.graph and hasattr(device, "capture"):
        graph, err = None, None
        try:
            graph = device.capture(order)
            device.replay(graph)                                # untimed: instantiation / first launch
            device.synchronize()
            if hasattr(device, "snapshot"):
                got = device.snapshot(order[-1])
                device.fwd(order[-1])
                device.bwd(order[-1])
                device.synchronize()
                if not device.same(got, device.snapshot(order[-1])):
                    raise RuntimeError("graph replay and eager launches disagree")
        except Exception as e:                                  # an extra measurement only
            err = f"{type(e).__name__}: {str(e)[:160]}"
            device.synchronize()
        # every rank takes the same path through the collectives below: a rank whose capture failed must not leave
        # the others waiting in a barrier (MAX over ranks of "failed")
        failed = max_over_ranks(1.0 if err else 0.0, nred, device.reduce_device()) > 0.0
        if failed:
            graph_replay = {"error": err or "graph capture failed on another rank"}
        else:
            barrier()
            t0 = time.perf_counter()
            device.replay(graph)
            barrier()
            eg = max_over_ranks(time.perf_counter() - t0, nred, device.reduce_device())
            graph_replay = {"ms_per_step": eg / K * 1e3, "value": whole_job_value(cnt["vox"], world, K, eg) / 1e9,
                            "note": "one hipGraph replay of the same K steps; not the metric"}

    # Extra, NOT the metric (--extras 1, LAB build of the library only -- csrc/Makefile `lab`): the same K steps with the
    # backward on the bf16 matrix pipe (every f32 operand split into three bf16 pieces: docs/lab_notebook.md 4.3).  The
    # product library does not contain that kernel (ABI 1.06); the field is null there.
    bf16x3 = None
    if args.extras and args.impl == 0 and hasattr(device, "bwd_as") and getattr(device, "is_lab_build", lambda: False)():
        def plain_pass(impl):
            """K eager steps between barriers, NO event records between the kernels (seconds, MAX over ranks)."""
            for i in range(min(K, 20)):
                device.fwd(order[i % K])
                device.bwd_as(order[i % K], impl)
            barrier()
            t0 = time.perf_counter()
            for z in order:
                device.fwd(z)
                device.bwd_as(z, impl)
            barrier()
            return max_over_ranks(time.perf_counter() - t0, nred, device.reduce_device())
        e0, e4 = plain_pass(0), plain_pass(4)
        bf16x3 = {"ms_per_step": e4 / K * 1e3, "value": whole_job_value(cnt["vox"], world, K, e4) / 1e9,
                  "default_same_method": {"ms_per_step": e0 / K * 1e3, "value": whole_job_value(cnt["vox"], world, K, e0) / 1e9},
                  "note": "lab build: K eager steps WITHOUT event records between the kernels, once with the bf16x3 backward (bf16 MFMA, "
                          "operands split in three) and once with the default backward; neither is the metric"}

    # Extra, NOT the metric (--ops 1, rank 0, a few seconds, after everything that is timed above): every other op / shape
    # of the path -- config 2, the three correlations the model really runs, ROIPool / PSROIPool at config 3 and at the
    # model's shapes -- through the C ABI with rotated buffers, each against its own roof (bench_ops.measure).
    ops = None
    if rank == 0 and args.ops and hasattr(device, "ops_extra"):
        try:
            ops = device.ops_extra()
        except Exception as e:                                  # an extra measurement only
            ops = {"error": f"{type(e).__name__}: {str(e)[:200]}"}

    if rank == 0:
        ms = elapsed / K * 1e3
        value = whole_job_value(cnt["vox"], world, K, elapsed) / 1e9

        def roof(name, us, nbytes, flops):
            t = sum(us) / len(us) * 1e-6                       # mean launch duration, s
            return dict(kernel=name, us=t * 1e6, us_min=min(us), us_max=max(us),
                        hbm=dict(achieved=nbytes / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=nbytes / t / 1e9 / HBM_PEAK_GBS),
                        mfma=dict(achieved=flops / t / 1e12, peak=FP32_MATRIX_PEAK_TF, unit="TFLOP/s",
                                  frac=flops / t / 1e12 / FP32_MATRIX_PEAK_TF))
        kernels = [roof("corr_fwd", us_fwd, cnt["fwd_bytes"], cnt["fwd_flops"]),
                   roof("corr_bwd", us_bwd, cnt["bwd_bytes"], cnt["bwd_flops"])]
        dom = max(kernels, key=lambda r: r["us"])
        # The correlation is f32-FMA-bound at this shape (AI 34-42 F/B vs a 19.7 F/B ridge,
        # SURVEY.md F10), so the binding roof of the dominant kernel is the f32 matrix/vector peak;
        # the HBM view BASELINE.json's metric asks for is reported beside it for every kernel.
        # `traffic` is NOT measured in this run: PMC counters need rocprofv3 around the process (separate --pmc passes,
        # tools/pmc_bwd.sh via tools/profile_round.sh).  It is read from the committed summary of those passes and the line
        # says which file / pass it came from (traffic_source).
        traffic, traffic_fwd, traffic_source = None, None, None
        tfile = ROOT / "profiles" / "traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                traffic = tj.get(args.workload, {}).get(dom["kernel"])
                traffic_fwd = tj.get(args.workload, {}).get("corr_fwd")
                traffic_source = {"file": "profiles/traffic.json", "measured_in_this_run": False, "pass": tj.get("_note")}
            except Exception:
                traffic, traffic_fwd, traffic_source = None, None, None
        t_dev = (kernels[0]["us"] + kernels[1]["us"]) * 1e-3
        ms_ev = elapsed_ev / K * 1e3
        if t_dev > ms_ev * 1.001:                               # the intervals tile the event pass: they cannot outlast it
            raise RuntimeError(f"inconsistent timing: kernels {t_dev:.4f} ms > event pass {ms_ev:.4f} ms per step")
        record_us = max(0.0, (ms_ev - ms) * 1e3 / 2.0)          # two records per step
        for k in kernels:
            k["us_minus_record_overhead"] = k["us"] - record_us
            # the launch itself: the event interval minus the one event record it contains (what rocprofv3's kernel duration agrees with)
            k["mfma"]["frac_launch"] = k["mfma"]["frac"] * k["us"] / k["us_minus_record_overhead"]
            k["hbm"]["frac_launch"] = k["hbm"]["frac"] * k["us"] / k["us_minus_record_overhead"]
        # Third roof: the L1 / texture-address line rate.  lines = cache-line (tag) accesses per launch
        # (TCP_TOTAL_CACHE_ACCESSES, PMC pass of the same command), peak = the chip-wide rate lab/csrc/ta_lab sustains with
        # every CU streaming whole lines; both from profiles/ta_roof.json, the duration measured here.
        ta = None
        tafile = ROOT / "profiles" / "ta_roof.json"
        if tafile.exists():
            try:
                ta = json.loads(tafile.read_text())
            except Exception:
                ta = None
        for k in kernels:
            k["roofline_ta"] = None
            lines = ((ta or {}).get(args.workload) or {}).get(k["kernel"])
            if ta and lines:
                k["roofline_ta"] = dict(lines=lines, peak=ta["peak_lines_per_s"], unit="lines/s", achieved=lines / (k["us"] * 1e-6),
                                        frac=lines / (k["us"] * 1e-6) / ta["peak_lines_per_s"], line_bytes=ta.get("line_bytes"),
                                        clock_ghz=ta.get("clock_ghz"))
        line = {
            "metric": METRIC,
            "value": value, "unit": "Gvox/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, **cfg, "per_gpu_batch": B, "global_batch": B * world,
                       "buffer_sets": n_sets, "impl": args.impl, "parallelism": f"shard{world}"},
            # frac: algorithmic FLOP / launch duration / peak, the launch duration being the HIP-event interval minus the one event
            # record it contains (launch_us); frac_event_interval: the same over the raw interval (launch_us_event_interval).
            "roofline": {"kernel": dom["kernel"], "bound": "mfma",
                         "achieved": dom["mfma"]["achieved"] * dom["us"] / dom["us_minus_record_overhead"],
                         "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s", "frac": dom["mfma"]["frac_launch"],
                         "frac_event_interval": dom["mfma"]["frac"],
                         "traffic": traffic, "traffic_source": traffic_source,
                         "launch_us": dom["us_minus_record_overhead"], "launch_us_event_interval": dom["us"],
                         "note": "bound 'mfma' = the exact-f32 matrix peak, which IS the f32 vector peak (157.3 TF/s): v_mfma_f32_16x16x4_f32 "
                                 "is bit for bit the reference's ascending-channel fmaf chain"},
            # the kernel BASELINE.json's metric string and north_star's target name (the forward), against both roofs
            "roofline_fwd": {"kernel": kernels[0]["kernel"], "launch_us": kernels[0]["us_minus_record_overhead"],
                             "launch_us_event_interval": kernels[0]["us"],
                             "pct_hbm": 100 * kernels[0]["hbm"]["frac_launch"], "pct_f32": 100 * kernels[0]["mfma"]["frac_launch"],
                             "hbm": {"achieved": kernels[0]["hbm"]["achieved"] * kernels[0]["us"] / kernels[0]["us_minus_record_overhead"],
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s"},
                             "f32": {"achieved": kernels[0]["mfma"]["achieved"] * kernels[0]["us"] / kernels[0]["us_minus_record_overhead"],
                                     "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s"},
                             "traffic": traffic_fwd, "traffic_source": traffic_source,
                             "target_pct_hbm": 70.0, "ceiling_pct_hbm_at_f32_peak": 57.4},
            "kernels": kernels,
            "fwd_gvox_per_s": cnt["vox"] / kernels[0]["us"] / 1e3, "bwd_gvox_per_s": cnt["vox"] / kernels[1]["us"] / 1e3,
            "pct_hbm_roofline_fwd": 100 * kernels[0]["hbm"]["frac"],
            "pct_hbm_roofline_bwd": 100 * kernels[1]["hbm"]["frac"],
            "settle_steps": settle_steps, "settle_ms": settle_ms, "ranks_seen": ranks_seen,
            "process_group": getattr(device, "backend", None) if dist_on else None,
            "gpus_visible": device.gpus_visible() if hasattr(device, "gpus_visible") else None,    # < n_gpus: a rehearsal, ranks share GPUs
            "event_pass": {"ms_per_step": ms_ev, "records_per_step": 2, "event_record_overhead_us": record_us,
                           "host_ms_per_step_minus_device": ms_ev - t_dev},
            "graph_replay": graph_replay,
            "bf16x3_backward": bf16x3,
            "ops": ops,
            "timing": {"value": "wall time of K eager steps between barriers, nothing else on the stream (ms_per_step)",
                       "kernels": "a second pass of the same K steps with a HIP event between the kernels: kernels[] / roofline are "
                                  "the event intervals, which tile event_pass.ms_per_step (asserted); an interval contains one "
                                  "event record (event_pass.event_record_overhead_us = the difference of the two passes per record)"},
        }
        if not args.no_cpu_baseline:
            if world == 1:
                line["cpu_baseline"] = cpu_baseline(cfg, cnt)
            else:
                # N > 1 (north_star: the CPU figure beside the 2 / 4 / 8-GPU lines): rank 0 alone, after everything timed, on its
                # share of the host's threads and a shorter sample -- the other ranks wait at the closing barrier meanwhile
                cb = cpu_baseline(cfg, cnt, budget_s=args.cpu_baseline_multi_s, threads=max(1, host_threads() // world))
                cb["sample"] += f" (rank 0 of {world}, 1/{world} of the host's threads; the per-GPU workload, not the {world}-GPU job)"
                line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)

    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    return dict(settle_steps=settle_steps, ranks_seen=ranks_seen)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="corr_B8_C256_38x63_d8", choices=sorted(WORKLOADS))
    ap.add_argument("--sets", type=int, default=0, help="rotated buffer sets (0 = enough for > 512 MiB)")
    ap.add_argument("--impl", type=int, default=0, help="0 auto, 1 generic kernels, 2 tuned only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-multi-s", type=float, default=5.0, help="N > 1: seconds of CPU sample on rank 0 (N = 1 uses ~12 s)")
    ap.add_argument("--graph", type=int, default=1, help="1: also replay the K steps as one HIP graph (reported beside the metric)")
    ap.add_argument("--extras", type=int, default=1, help="1: with a LAB build of the library also time the K steps with the bf16x3 backward (reported beside the metric)")
    ap.add_argument("--ops", type=int, default=1, help="1: also time every other op / shape of the path (ops[] in the line, rank 0, ~3 s)")
    ap.add_argument("--force-dist", type=int, default=0, help="1: create the process group and run the collectives at world size 1 too "
                    "(launch under torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--backend", default="nccl", help="process-group backend: nccl = RCCL, one GPU per rank (the metric); gloo lets several "
                    "ranks REHEARSE the N > 1 path on one GPU (they share it: the line says so, its value is not a scaling figure)")
    ap.add_argument("--settle-ms", type=float, default=SETTLE_MS, help="untimed steps until this much wall time has passed (clock settling)")
    return ap.parse_args(argv)


def main(argv=None, device_factory=None, script=None):
    """`python3 bench.py --gpus N ...`.  Under torch.distributed.run (WORLD_SIZE set) this process is one rank.  Launched
    bare with N > 1 it is the GPU-free parent that starts the N ranks (launch_ranks) and exits with their code.
    device_factory / script: the test harness (tests/bench_stub_main.py) runs the same entry with its own device layer."""
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(script or Path(__file__).resolve(), argv, args.gpus))
    device = device_factory() if device_factory else HipDevice(rank_env()[2], args.impl, args.backend)
    return run(args, device)


if __name__ == "__main__":
    main()
