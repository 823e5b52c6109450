"""Child program of tests/test_data_parallel_equivalence.py (not a test module itself).

The reference sums the pair losses of a minibatch, divides the scalar by the pair count and runs ONE backward
(/root/reference/detect_to_track/trainer.py:258-276, utils.py:64-75); SURVEY §8e says that mean-reducing the gradients of
ranks with equal shards reproduces the single-process gradient over all pairs.  This program produces both sides of that
statement on the GPU with the REAL model (ResNet-50 + RPN + R-FCN + correlation tracker, exact tracker forward) and the real
training step (detect_to_track/training.py: DataParallelTrainer.train_step):

    --mode single --out F     one process, ALL 2P pairs in one minibatch, no process group; writes gradients + losses to F
    --mode ranks  --ref F     under torch.distributed.run with W ranks (gloo; the ranks may share one GPU): every rank runs
                              train_step on ITS shard of the same 2P pairs (BatchLoader: rank r keeps positions r, r + W, ...),
                              GradientBuckets averages the gradients; every rank compares ALL trainable parameters with F.

Both modes use the reference's Python loop over pairs (``batched=False``): every library / MIOpen call then has the same
shape in both runs, so the per-pair arithmetic is the same and only the order of the sum over pairs differs.
"""
import argparse
import json
import os
import sys
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))

H, W, R, T = 320, 480, 64, 4                     # frame, regions per frame, tracked boxes per pair
COEFS = [1.0, 1.0, 1.0, 1.0, 1.0e-4]             # cfg/default.yaml COEFS


def build(dev, buckets_factory=None):
    from detect_to_track.models import DetectTrackModule
    from detect_to_track.training import DataParallelTrainer, RegionProposals, build_anchors
    # MIOpen / rocBLAS may pick kernels that add partial sums with atomics (GEMM-based convolutions, split-K): the step is then not even
    # reproducible between two runs of the SAME process layout, and a rounding-level difference in the RPN outputs can flip a region
    # (top-k / NMS / a bin's floor or ceil) -- measured: 1e-2 of a parameter's largest gradient between two identical single-process runs,
    # lab/tools/step_determinism.py.  Ask for deterministic library kernels, so that what is compared is the data-parallel arithmetic.
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    torch.manual_seed(0)                         # the SAME initial weights in every process
    model = DetectTrackModule("resnet50", 3, 15, 30, 7, 8, 7).to(dev).train()
    assert not model.c_tracker.fast_forward      # the exact tracker forward (bit-identical to the reference's kernels)
    params = [p for p in model.parameters() if p.requires_grad]
    with torch.no_grad():
        fh, fw = model.backbone(torch.rand(1, 3, H, W, device=dev))["c4"].shape[-2:]
    anchors = build_anchors((fh, fw), [0.001, 0.004, 0.016, 0.064, 0.256], [0.5, 1.0, 2.0])
    optim = torch.optim.SGD(params, lr=0.0)      # lr 0, no momentum / decay: the step leaves weights AND p.grad as they are
    buckets = buckets_factory(params) if buckets_factory else None
    trainer = DataParallelTrainer(model, optim, torch.tensor(COEFS, device=dev), RegionProposals(anchors, 0.3, R, 0.5, dev),
                                  buckets, batched=False)
    return model, trainer, len(anchors)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("single", "ranks"), required=True)
    ap.add_argument("--pairs-per-rank", type=int, default=2)
    ap.add_argument("--world", type=int, default=2, help="ranks of the data-parallel run (both modes must agree)")
    ap.add_argument("--out")
    ap.add_argument("--ref")
    ap.add_argument("--rtol", type=float, default=1e-5)
    a = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X: the ops have no CPU path"
    from detect_to_track.training import BatchLoader, SyntheticPairManager
    n_pairs = a.world * a.pairs_per_rank
    if a.mode == "single":
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        model, trainer, n_anchor = build(dev)
        manager = SyntheticPairManager(n_pairs, (H, W), n_anchor, R, T, 30, dev, seed=11)
        minibatch = next(iter(BatchLoader(manager, n_pairs, 0, 1, seed=5)))
        assert len(minibatch) == n_pairs
        total, _, _ = trainer.train_step(minibatch)
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
        assert all(torch.isfinite(g).all() for g in grads.values()) and any(bool(g.any()) for g in grads.values())
        torch.save({"grads": grads, "loss_mean": (total.detach() / n_pairs).cpu(), "pairs": n_pairs}, a.out)
        print(json.dumps({"mode": "single", "pairs": n_pairs, "params": len(grads)}))
        return

    import torch.distributed as dist
    from detect_to_track.data_parallel import GradientBuckets
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.world
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    try:
        model, trainer, n_anchor = build(dev, lambda params: GradientBuckets(params, bucket_mb=16.0))
        assert len(trainer.buckets.buckets) >= 3                                   # several buckets, several collectives
        manager = SyntheticPairManager(n_pairs, (H, W), n_anchor, R, T, 30, dev, seed=11)
        minibatch = next(iter(BatchLoader(manager, a.pairs_per_rank, rank, world, seed=5)))   # this rank's disjoint shard
        assert len(minibatch) == a.pairs_per_rank
        total, _, _ = trainer.train_step(minibatch)
        torch.cuda.synchronize()
        ref = torch.load(a.ref)
        assert ref["pairs"] == n_pairs
        loss_mean = (total.detach() / a.pairs_per_rank).cpu()
        dist.all_reduce(loss_mean)                                                  # gloo, CPU tensor
        loss_mean /= world
        worst, worst_name, n, table = 0.0, None, 0, []
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            g, want = p.grad.detach().cpu(), ref["grads"][name]
            scale = float(want.abs().max())
            err = float((g - want).abs().max()) / max(scale, 1e-30)
            table.append((err, name, scale, tuple(want.shape)))
            if err > worst:
                worst, worst_name = err, name
            n += 1
        if rank == 0:
            for err, name, scale, shape in sorted(table, reverse=True)[:8]:
                print(f"[dp_equivalence] {name} {shape}: max|diff| / max|grad| = {err:.3e} (max|grad| {scale:.3e})", file=sys.stderr, flush=True)
        loss_err = float(((loss_mean - ref["loss_mean"]).abs() / ref["loss_mean"].abs().clamp_min(1e-30)).max())
        ok = worst <= a.rtol and loss_err <= a.rtol and n == len(ref["grads"])
        flag = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                                 # EVERY rank holds the single-process gradient
        if rank == 0:
            print(json.dumps({"mode": "ranks", "world": world, "pairs": n_pairs, "params": n, "worst_rel_err": worst, "worst_param": worst_name,
                              "loss_rel_err": loss_err, "all_ranks_ok": bool(flag.item() == 1.0), "buckets": len(trainer.buckets.buckets)}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
