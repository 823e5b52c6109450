#!/usr/bin/env python3
"""bench_model.py -- BASELINE.json config 4: one training step of the full DetectTrack graph on ONE MI355X.

What is timed: the reference's training step (trainer.py:133-256, 270-281) for a minibatch of B frame
pairs -- backbone on the two frames of a pair, RPN, R-FCN on both frames' regions, correlation tracker,
the five losses summed over the minibatch, ONE backward, one SGD step -- with random weights and synthetic
frames (there is no network for datasets or checkpoints).  The step itself is detect_to_track/training.py
(DataParallelTrainer over a SyntheticPairManager): region proposals are decoded, filtered and NMS-ed ON THE DEVICE
from the live RPN outputs (csrc/d2t_regions.hip -- the reference does this on host copies, trainer.py:178-207:
no device->host copy is left between the RPN and R-FCN); what is NOT the reference's are the host-side label
encoders (numpy + third-party code, SURVEY §2 rows 13-15: synthetic targets of the same shapes) and the loss classes
(plain smooth-L1 / NLL terms); the ops, their shapes and their call pattern are the reference's.

The deliverable is the step-time breakdown: which part of a step the three custom ops
(PointwiseCorrelation, ROIPool, PSROIPool) are.  Every call into the HIP library is bracketed by HIP
events on torch's current stream (forward and backward alike: both go through ``_ext``), so the ops'
device time is measured inside the running step, not in isolation.

Input shape: BASELINE.json says 3x600x1000, but the reference's own tracker cannot run that shape --
c3 (75x125 at stride 8) halves to 37x62 (correlation_tracker.py:60-61) while c4/c5 are 38x63, and the
`torch.cat` at :72 fails.  The default here is the nearest shape the reference accepts, 608x1008
(c4 = 38x63, the BASELINE correlation shape); `--height 608 --width 1200` is the reference's
cfg/default.yaml INPUT_SHAPE.

Config 5 (same model, B pairs PER GPU, gradients averaged over RCCL): `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 bench_model.py --gpus N` -- detect_to_track/data_parallel.py packs the 150 MB
of gradients into 64 MB buckets and launches each all-reduce from autograd's hooks, under the rest of the backward
pass.  That path is covered on CPU/gloo (tests/test_data_parallel_gloo.py) and has NOT run on multi-GPU hardware.

One JSON line on stdout (rank 0).  Not the headline metric (that is bench.py).
"""
import argparse
import json
import os
import sys
import time

# The host driver of this pool only supports dmabuf IPC: without this RCCL (and CUDA-tensor sharing across processes) fails
# with "hipIpcGetMemHandle: invalid argument".  Must be in the environment before HIP initialises; harmless elsewhere.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from collections import OrderedDict, defaultdict
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--height", type=int, default=608)
    p.add_argument("--width", type=int, default=1008)
    p.add_argument("--pairs", type=int, default=2, help="frame pairs per minibatch (BASELINE config 4: 2)")
    p.add_argument("--rois", type=int, default=300, help="regions per frame handed to R-FCN (BASELINE config 3: 300)")
    p.add_argument("--track-rois", type=int, default=8, help="tracked objects per pair")
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--backbone", default="resnet50")
    p.add_argument("--lr", type=float, default=1e-5)
    p.add_argument("--gpus", type=int, default=1, help="ranks (BASELINE config 5); launch with torch.distributed.run, one rank per GPU")
    p.add_argument("--bucket-mb", type=float, default=64.0, help="gradient all-reduce bucket size")
    p.add_argument("--backend", default="nccl", help="nccl = RCCL (one GPU per rank); gloo lets several ranks rehearse on ONE GPU")
    p.add_argument("--force-dist", action="store_true", help="create the process group, the gradient buckets and their all-reduces at world size 1 "
                   "too (under torch.distributed.run --nproc-per-node 1): exercises the RCCL path on a box with one GPU")
    p.add_argument("--per-pair", action="store_true", help="the reference's Python loop over the pairs of a minibatch (B = 1 op calls) instead of "
                   "one batched pass (training.py: forward_loss_pairs)")
    p.add_argument("--fast-tracker", action="store_true", help="opt in to D2T_IMPL_FAST for the tracker's correlation forward (channel split, within 1e-5, "
                   "NOT bit-identical).  Default: the reference's arithmetic -- the exact forward (ADVICE round 4: the config-4 line must stay comparable)")
    p.add_argument("--exact-tracker", action="store_true", help="(default since round 5; kept so that older command lines still parse)")
    p.add_argument("--miopen-find", action="store_true", help="let MIOpen benchmark its convolution algorithms (torch.backends.cudnn.benchmark): "
                   "a long first step, faster library convolutions afterwards")
    return p.parse_args(argv)


class OpTimer:
    """Brackets every tensor-level entry point of the HIP library with events on the current stream."""
    NAMES = ("pointwise_correlation_forward", "pointwise_correlation_backward",
             "pointwise_correlation_levels_forward", "pointwise_correlation_levels_backward",
             "roipool_forward", "roipool_backward", "ps_roipool_forward", "ps_roipool_backward", "region_filter", "region_filter_batched")

    def __init__(self, ext):
        self.pending = []
        self.enabled = False
        for name in self.NAMES:
            setattr(ext, name, self._wrap(name, getattr(ext, name)))

    def _wrap(self, name, fn):
        def timed(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.pending.append((name, e0, e1))
            return out
        return timed

    def collect(self):
        acc = defaultdict(float)
        calls = defaultdict(int)
        for name, e0, e1 in self.pending:
            acc[name] += e0.elapsed_time(e1)
            calls[name] += 1
        self.pending = []
        return acc, calls


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare (the way the driver launches N = 1): this process starts the N ranks as a child
        # `python -m torch.distributed.run ...` BEFORE any torch.cuda / HIP call and exits with their code (bench.launch_ranks)
        sys.path.insert(0, str(ROOT))
        import bench
        raise SystemExit(bench.launch_ranks(Path(__file__).resolve(), argv, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench_model.py needs an MI355X: the ops have no CPU path")
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: main() starts the ranks itself when launched bare")
    from detect_to_track.models import DetectTrackModule, _ext
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist_on = world > 1 or args.force_dist                                         # --force-dist: the RCCL path at world size 1 (one-GPU rehearsal)
    buckets = None
    if dist_on:                                                                    # config 5: weak scaling, B pairs per GPU,
        import torch.distributed as dist                                            # gradients averaged over RCCL / xGMI
        from detect_to_track.data_parallel import GradientBuckets
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    if args.miopen_find:
        torch.backends.cudnn.benchmark = True
        import threading
        stop = threading.Event()

        def heartbeat():                                                           # the algorithm search is silent for minutes
            t0 = time.time()
            while not stop.wait(45):
                print(f"[bench_model] MIOpen find in progress, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        threading.Thread(target=heartbeat, daemon=True).start()
    torch.manual_seed(0)                                                           # identical initial weights on every rank
    timer = OpTimer(_ext)
    # cfg/default.yaml: resnet50, first trainable stage 3, 5 areas x 3 ratios = 15 anchors, 30 classes, k = 7, d_max = 8
    model = DetectTrackModule(args.backbone, 3, 15, 30, 7, 8, 7).to(dev)
    model.c_tracker.fast_forward = bool(args.fast_tracker) and not args.exact_tracker   # D2T_IMPL_FAST is an opt-in; the default forward is bit-identical to the reference
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    # cfg SGD_KWARGS with the learning rate turned down: random weights against random targets diverge at 1e-2 within two
    # steps, and non-finite feature maps would put the ops on their (slow, cold) non-finite repair paths
    optim = torch.optim.SGD(params, lr=args.lr, weight_decay=1e-4, momentum=0.9)
    coefs = torch.tensor([1.0, 1.0, 1.0, 1.0, 1.0e-4], device=dev)           # cfg COEFS
    if dist_on:
        buckets = GradientBuckets(params, bucket_mb=args.bucket_mb)
    torch.manual_seed(1 + rank)                                                    # every rank its own frames

    H, W, B, R, Rt = args.height, args.width, args.pairs, args.rois, args.track_rois
    from detect_to_track.training import BatchLoader, DataParallelTrainer, RegionProposals, SyntheticPairManager, build_anchors
    with torch.no_grad():
        fh, fw = model.backbone(torch.rand(1, 3, H, W, device=dev))["c4"].shape[-2:]
    anchors = build_anchors((fh, fw), [0.001, 0.004, 0.016, 0.064, 0.256], [0.5, 1.0, 2.0])     # cfg/default.yaml:13-14
    n_anchor = len(anchors)
    # cfg/default.yaml:21-23: confidence 0.3, NMS IoU 0.5; the list is capped at R regions per frame (BASELINE config 3: 300)
    regions = RegionProposals(anchors, 0.3, R, 0.5, dev)
    n_steps = args.warmup + args.steps
    manager = SyntheticPairManager(world * B * n_steps, (H, W), n_anchor, R, Rt, 30, dev, seed=1)
    loader = iter(BatchLoader(manager, B, rank, world, seed=0))                  # every rank its own pairs
    # Synthetic frames and targets are generated (and the tracked boxes copied host->device, a blocking copy) when a pair is
    # indexed: draw every minibatch of the run BEFORE anything is timed, so that step_ms holds the training step only.
    batches = iter([next(loader) for _ in range(n_steps)])
    torch.cuda.synchronize()
    trainer = DataParallelTrainer(model, optim, coefs, regions, buckets, batched=not args.per_pair)
    sections = DataParallelTrainer.SECTIONS

    def step(record):
        """One minibatch (training.py: forward of every pair, one backward, gradient all-reduce, one optimizer step)."""
        def mark():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        total, stamps, (b0e, b1e, b2e) = trainer.train_step(next(batches), mark)
        if record is not None:
            record.append((stamps, b0e, b1e, b2e))
        return total

    if rank == 0:
        print(f"[bench_model] {world} rank(s) x {B} pairs of 3x{H}x{W}, c4 {fh}x{fw}, {R} regions per frame, {Rt} tracked boxes; "
              f"warming up (first MIOpen calls compile kernels)", file=sys.stderr, flush=True)
    for w in range(args.warmup):
        t = time.time()
        step(None)
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[bench_model] warmup step {w}: {time.time() - t:.1f} s", file=sys.stderr, flush=True)
    if dist_on:
        dist.barrier()

    timer.enabled = True
    rec = []
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(args.steps):
        step(rec)
    s1.record()
    torch.cuda.synchronize()
    step_ms = s0.elapsed_time(s1) / args.steps
    if dist_on:                                                                    # the slowest rank defines the step
        t = torch.tensor([step_ms], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        step_ms = float(t.item())
    with torch.no_grad():                                                         # the ops' cold non-finite paths must not be what was timed
        fm = model.backbone(manager[0].frames)
        amax = {k: float(v.abs().max()) for k, v in fm.items()}
    finite = all(np.isfinite(v) for v in amax.values()) and all(bool(torch.isfinite(p).all()) for p in params)
    if rank == 0:
        print(f"[bench_model] after the run: max |activation| {amax}, parameters finite: {finite}", file=sys.stderr, flush=True)
    sec = defaultdict(float)
    for ev, b0e, b1e, b2e in rec:
        for marks in ev:
            for k, name in enumerate(sections):
                sec[name] += marks[k].elapsed_time(marks[k + 1])
        sec["backward"] += b0e.elapsed_time(b1e)
        sec["optimizer"] += b1e.elapsed_time(b2e)
    sec = {k: v / args.steps for k, v in sec.items()}
    ops, calls = timer.collect()
    ops = {k: v / args.steps for k, v in ops.items()}
    calls = {k: v // args.steps for k, v in calls.items()}
    fam = {"correlation": sum(v for k, v in ops.items() if k.startswith("pointwise")),
           "roipool": sum(v for k, v in ops.items() if k.startswith("roipool")),
           "ps_roipool": sum(v for k, v in ops.items() if k.startswith("ps_roipool")),
           "region_filter": ops.get("region_filter", 0.0) + ops.get("region_filter_batched", 0.0)}
    ops_ms = sum(fam.values())
    line = {
        "bench": "DetectTrack training step (BASELINE config %d)" % (4 if world == 1 else 5), "n_gpus": world, "scaling": "weak",
        "parallelism": f"dp{world}" if world > 1 else "single", "dtype": "f32", "data": "synthetic",
        "weights": "random", "process_group": args.backend if dist_on else None, "gradient_buckets": buckets is not None, "pairs_batched": not args.per_pair, "tracker_fast_forward": bool(args.fast_tracker) and not args.exact_tracker, "steps": args.steps, "warmup": args.warmup, "miopen_find": bool(args.miopen_find),
        "config": {"workload": f"detecttrack_{args.backbone}_B{B}pairs_3x{H}x{W}", "pairs": B, "frame": [3, H, W],
                   "c4": [fh, fw], "regions_per_frame": R, "tracked_boxes": Rt, "anchors": n_anchor},
        "ms_per_step": step_ms, "pairs_per_s": world * B / step_ms * 1e3, "pairs_per_gpu": B, "finite": finite, "max_abs_activation": amax,
        "sections_ms": sec,
        "custom_ops_ms": ops, "custom_ops_calls_per_step": calls, "custom_ops_by_family_ms": fam,
        "custom_ops_ms_total": ops_ms, "custom_ops_frac_of_step": ops_ms / step_ms,
        "regions": "device (d2t_region_filter_f32 on the live RPN outputs: no device->host copy between RPN and R-FCN)",
        "note": "sections are forward parts per step (all pairs); custom_ops_* are HIP-event brackets around every call "
                "into libd2t_ops.so (forward and backward), measured inside the running step",
    }
    if rank == 0:
        print(json.dumps(line))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
