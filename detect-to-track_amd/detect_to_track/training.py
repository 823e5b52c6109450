"""The reference's training step, sharded over ranks (BASELINE configs 4 and 5).

What of the reference is mirrored here, and where it differs:

* ``DataManager`` / ``BatchLoader`` -- the iteration protocol of ``detect_to_track/data/types.py:44-68`` and
  ``trainer.py:30-42`` (index a manager, get ``(instance_0, instance_1)`` frame pairs, group them into
  minibatches).  ``SyntheticPairManager`` stands in for the ImageNet VID/DET managers (``data/imagenet.py``: dataset
  I/O, out of scope): seeded synthetic frames and targets of the model's shapes, generated on the device.
  ``BatchLoader`` adds the one thing config 5 needs: every rank walks its OWN disjoint share of each epoch's
  permutation (rank r takes pairs r, r + world, ... -- "one pair per rank" of north_star, two per rank for config 5).
* ``RegionProposals`` -- the RPN -> R-FCN hand-over of ``trainer.py:178-207`` / ``inference.py:78-91`` (numpy
  ``frcnn_box_decode`` + three ``ml_utils`` filters on host copies) as ONE device call, ``_ext.region_filter``
  (csrc/d2t_regions.hip): no device->host->device round trip, static shapes (padded to ``max_dets``).
* ``DataParallelTrainer.forward_loss`` follows ``DetectTrackTrainer._forward_loss`` (``trainer.py:133-256``): backbone on
  the two frames of a pair, RPN, regions, R-FCN on both frames, correlation tracker, five loss terms.  The
  reference's label ENCODERS (``data/encoding.py``, numpy + ``ml_utils``) and loss classes (``loss.py``: focal /
  masked smooth-L1) are out of scope: targets come from the manager and the terms are plain
  NLL / smooth-L1 of the same shapes -- parity for this row is about shapes, call pattern and the ops inside.
* ``DataParallelTrainer.train_step`` follows ``_minibatch_loss`` + ``train`` (``trainer.py:258-281``): losses summed over
  the minibatch's pairs, ONE backward of ``dot(coefs, losses) / pairs`` (``DTLoss.to_scalar``, ``utils.py:64-75``: the
  scalar is normalised by the pair count), one optimizer step -- plus, between the two, the bucketed gradient all-reduce
  of ``data_parallel.GradientBuckets`` when there is more than one rank.  Because every rank normalises by ITS pair
  count, the mean over ranks with equal shards IS the single-process gradient over all pairs (SURVEY §8e;
  tests/test_data_parallel_equivalence.py checks that on the GPU with the real model).
"""
from collections import OrderedDict
from typing import Callable, Iterator, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor


def build_anchors(fm_shape: Tuple[int, int], anchor_areas: Sequence[float], aspect_ratios: Sequence[float]) -> np.ndarray:
    """(H * W * |areas x ratios|, 4) anchor boxes (centre_i, centre_j, h, w) as fractions -- reference utils.py:114-163."""
    dims = np.array([[np.sqrt(a * r), a / np.sqrt(a * r)] for a in anchor_areas for r in aspect_ratios])
    fm_h, fm_w = fm_shape
    iv, jv = np.meshgrid(np.linspace(0, 1, fm_h, endpoint=False) + 1 / fm_h / 2,
                         np.linspace(0, 1, fm_w, endpoint=False) + 1 / fm_w / 2, indexing="ij")
    ij = np.broadcast_to(np.stack([iv, jv], axis=-1)[:, :, None, :], (fm_h, fm_w, len(dims), 2))
    hw = np.broadcast_to(dims[None, None, :, :], (fm_h, fm_w, len(dims), 2))
    return np.concatenate([ij, hw], 3).reshape(-1, 4).astype(np.float32)


class PairInstance(NamedTuple):
    """One training example: two consecutive frames and the encoded targets of the model's heads."""
    frames: Tensor        # (2, 3, H, W)
    o_star: Tensor        # (2, |A|) int64: anchor is object / background            (RPN classification)
    b_star: Tensor        # (2, |A|, 4): anchor offsets                               (RPN regression)
    c_star: Tensor        # (2 * R,) int64 in [0, n_classes]: region classes          (R-FCN classification)
    r_star: Tensor        # (2 * R, 4): region offsets                                (R-FCN regression)
    track_rois: Tensor    # (T, 4) boxes present in both frames, inside the frame     (tracker input, trainer.py:238)
    t_star: Tensor        # (T, 4): cross-frame offsets                               (tracker regression)


def inside_rois(n: int, rng: np.random.Generator) -> np.ndarray:
    """(n, 4) ijhw fractions INSIDE the frame: ROIPool gives 0/0 = NaN for a bin that lies outside the map (like the
    reference, roipool_cuda.cu:61), and the trainer's tracked boxes are ground-truth boxes."""
    ctr = rng.uniform(0.15, 0.85, (n, 2))
    size = np.minimum(rng.uniform(0.05, 0.6, (n, 2)), 1.9 * np.minimum(ctr, 1.0 - ctr))
    return np.concatenate([ctr, size], 1).astype(np.float32)


class SyntheticPairManager:
    """A ``DataManager`` (reference data/types.py:44-55: ``__getitem__`` + ``__len__``) of seeded synthetic pairs.
    Pair i is a pure function of (seed, i): every rank can index any pair and gets the same tensors."""

    def __init__(self, length: int, frame_hw: Tuple[int, int], n_anchors: int, regions_per_frame: int, tracked: int,
                 n_classes: int, device: torch.device, seed: int = 0) -> None:
        self.length, self.hw, self.A, self.R, self.T = int(length), tuple(frame_hw), int(n_anchors), int(regions_per_frame), int(tracked)
        self.n_classes, self.device, self.seed = int(n_classes), device, int(seed)

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, i: int) -> PairInstance:
        if not 0 <= i < self.length:
            raise IndexError(i)
        g = torch.Generator(device=self.device).manual_seed(self.seed * 1_000_003 + i)
        rnd = dict(device=self.device, generator=g)
        H, W = self.hw
        return PairInstance(
            frames=torch.rand(2, 3, H, W, **rnd),
            o_star=torch.randint(0, 2, (2, self.A), **rnd),
            b_star=torch.randn(2, self.A, 4, **rnd),
            c_star=torch.randint(0, self.n_classes + 1, (2 * self.R,), **rnd),
            r_star=torch.randn(2 * self.R, 4, **rnd),
            track_rois=torch.from_numpy(inside_rois(self.T, np.random.default_rng(self.seed * 7919 + i))).to(self.device),
            t_star=torch.randn(self.T, 4, **rnd),
        )


class BatchLoader:
    """Minibatches of ``batch_size`` pairs for THIS rank (reference trainer.py:30-42: random order, drop_last).  All
    ranks draw the same permutation of an epoch (seeded) and rank r keeps positions r, r + world, ...: disjoint
    shards, no communication."""

    def __init__(self, manager, batch_size: int, rank: int = 0, world: int = 1, seed: int = 0) -> None:
        self.manager, self.batch_size, self.rank, self.world, self.seed = manager, int(batch_size), int(rank), int(world), int(seed)
        self.epoch = 0

    def __len__(self) -> int:
        return (len(self.manager) // self.world) // self.batch_size

    def __iter__(self) -> Iterator[List[PairInstance]]:
        order = np.random.default_rng(self.seed + self.epoch).permutation(len(self.manager))
        self.epoch += 1
        mine = order[self.rank::self.world][: len(self) * self.batch_size]
        for k in range(0, len(mine), self.batch_size):
            yield [self.manager[int(j)] for j in mine[k: k + self.batch_size]]


class RegionProposals:
    """RPN outputs of one frame -> (max_dets, 4) region boxes on the device (trainer.py:98-102,178-190)."""

    def __init__(self, anchors: np.ndarray, conf_thresh: float, max_dets: int, iou_thresh: float, device: torch.device) -> None:
        self.anchors = torch.from_numpy(np.ascontiguousarray(anchors, dtype=np.float32)).to(device)
        self.conf_thresh, self.max_dets, self.iou_thresh = float(conf_thresh), int(max_dets), float(iou_thresh)

    @torch.no_grad()
    def __call__(self, obj_conf: Tensor, offsets: Tensor) -> Tuple[Tensor, Tensor]:
        """obj_conf (|A|,) confidence of "object" (trainer.py:179-182), offsets (|A|, 4) -> (boxes, count)."""
        from .models import _ext
        boxes, _, _, count = _ext.region_filter(self.anchors, offsets.detach().contiguous(), obj_conf.detach().contiguous(),
                                                self.conf_thresh, self.max_dets, self.iou_thresh)
        return boxes, count

    @torch.no_grad()
    def frames(self, obj_conf: Tensor, offsets: Tensor) -> Tuple[Tensor, Tensor]:
        """All N frames of a step in ONE library call: obj_conf (N, |A|), offsets (N, |A|, 4) -> (boxes (N, max_dets, 4), counts (N,))."""
        from .models import _ext
        boxes, _, _, count = _ext.region_filter_batched(self.anchors, offsets.detach().contiguous(), obj_conf.detach().contiguous(),
                                                        self.conf_thresh, self.max_dets, self.iou_thresh)
        return boxes, count


class DataParallelTrainer:
    """One rank of the training job.  ``buckets`` is a ``data_parallel.GradientBuckets`` over the trainable parameters
    (None for a single process)."""

    SECTIONS = ("backbone", "rpn", "regions", "rcnn", "tracker", "loss")

    def __init__(self, model, optimizer, loss_coefs: Tensor, regions: RegionProposals, buckets=None, batched: bool = False) -> None:
        """batched: run the P pairs of a minibatch through the model TOGETHER (forward_loss_pairs) instead of the reference's
        Python loop over pairs (forward_loss): one backbone / RPN pass over the 2P frames, ONE region-filter call for all
        frames, ONE fused correlation call with B = P for the tracker -- the B = 1 calls of the loop leave most of the chip
        idle.  Same losses per pair; the convolutions see another batch size, so values agree to rounding, not bit for bit."""
        self.model, self.optim, self.coefs, self.regions, self.buckets, self.batched = model, optimizer, loss_coefs, regions, buckets, bool(batched)

    def forward_loss(self, inst: PairInstance, mark: Optional[Callable[[], object]] = None) -> Tuple[Tensor, list]:
        """The five loss terms of one pair (reference trainer.py:133-256) and, with ``mark``, the section boundaries."""
        m = self.model
        stamps = [mark()] if mark else []
        fmaps = m.backbone(inst.frames)                                           # trainer.py:152
        stamps += [mark()] if mark else []
        o_hat, b_hat, fm_reg = m.rpn(fmaps["c4"])                                 # :164
        stamps += [mark()] if mark else []
        rboxes_0, _ = self.regions(o_hat[0, :, 1], b_hat[0])                      # :178-190, on the device
        rboxes_1, _ = self.regions(o_hat[1, :, 1], b_hat[1])
        stamps += [mark()] if mark else []
        c5_0, c5_1 = fmaps["c5"]
        c0, b0 = m.rcnn(c5_0, rboxes_0)                                           # :207-208
        c1, b1 = m.rcnn(c5_1, rboxes_1)
        stamps += [mark()] if mark else []
        pyr0 = OrderedDict((k, fmaps[k][0]) for k in ("c3", "c4", "c5"))
        pyr1 = OrderedDict((k, fmaps[k][1]) for k in ("c3", "c4", "c5"))
        t_hat = m.c_tracker(pyr0, pyr1, fm_reg[0], fm_reg[1], inst.track_rois)    # :238
        stamps += [mark()] if mark else []
        nll, sl1 = torch.nn.functional.nll_loss, torch.nn.functional.smooth_l1_loss
        c_hat, r_hat = torch.cat([c0, c1]), torch.cat([b0, b1])
        losses = torch.stack([
            nll(torch.log(o_hat.reshape(-1, 2) + 1e-8), inst.o_star.reshape(-1)),
            sl1(b_hat, inst.b_star),
            nll(torch.log(c_hat + 1e-8), inst.c_star),
            sl1(r_hat, inst.r_star),
            sl1(t_hat, inst.t_star),
        ])
        stamps += [mark()] if mark else []
        return losses, stamps

    def forward_loss_pairs(self, minibatch: Sequence[PairInstance], mark: Optional[Callable[[], object]] = None) -> Tuple[Tensor, list]:
        """forward_loss for the P pairs of a minibatch at once (frames interleaved: pair p = frames 2p, 2p+1).  Returns the
        SUM over pairs of the five loss terms and one list of section stamps."""
        m, P = self.model, len(minibatch)
        stamps = [mark()] if mark else []
        fmaps = m.backbone(torch.cat([inst.frames for inst in minibatch]))        # (2P, 3, H, W)
        stamps += [mark()] if mark else []
        o_hat, b_hat, fm_reg = m.rpn(fmaps["c4"])                                 # (2P, |A|, 2), (2P, |A|, 4), (2P, Cr, h, w)
        stamps += [mark()] if mark else []
        rboxes, _ = self.regions.frames(o_hat[:, :, 1], b_hat)                    # one call for the 2P frames
        stamps += [mark()] if mark else []
        cls, reg = [], []
        for f in range(2 * P):                                                    # R-FCN pools one map at a time (rfcn.py:66-84)
            c, b = m.rcnn(fmaps["c5"][f], rboxes[f])
            cls.append(c)
            reg.append(b)
        stamps += [mark()] if mark else []
        pyr0 = OrderedDict((k, fmaps[k][0::2]) for k in ("c3", "c4", "c5"))
        pyr1 = OrderedDict((k, fmaps[k][1::2]) for k in ("c3", "c4", "c5"))
        t_hats = m.c_tracker.forward_pairs(pyr0, pyr1, fm_reg[0::2], fm_reg[1::2], [inst.track_rois for inst in minibatch])
        stamps += [mark()] if mark else []
        nll, sl1 = torch.nn.functional.nll_loss, torch.nn.functional.smooth_l1_loss
        total = torch.zeros(5, device=self.coefs.device)
        for p, inst in enumerate(minibatch):
            o_p, b_p = o_hat[2 * p: 2 * p + 2], b_hat[2 * p: 2 * p + 2]
            c_hat, r_hat = torch.cat(cls[2 * p: 2 * p + 2]), torch.cat(reg[2 * p: 2 * p + 2])
            total = total + torch.stack([
                nll(torch.log(o_p.reshape(-1, 2) + 1e-8), inst.o_star.reshape(-1)),
                sl1(b_p, inst.b_star),
                nll(torch.log(c_hat + 1e-8), inst.c_star),
                sl1(r_hat, inst.r_star),
                sl1(t_hats[p], inst.t_star),
            ])
        stamps += [mark()] if mark else []
        return total, stamps

    def train_step(self, minibatch: Sequence[PairInstance], mark: Optional[Callable[[], object]] = None):
        """trainer.py:258-281 for one minibatch.  Returns (summed losses, per-pair section stamps, (b0, b1, b2) marks
        around backward + all-reduce and the optimizer step).  The gradient is that of the MEAN over the minibatch's pairs."""
        total = torch.zeros(5, device=self.coefs.device)
        stamps = []
        if self.batched:
            total, st = self.forward_loss_pairs(minibatch, mark)
            stamps.append(st)
        else:
            for inst in minibatch:                                                # the reference's Python loop over pairs (:263-264)
                losses, st = self.forward_loss(inst, mark)
                total = total + losses
                stamps.append(st)
        if self.buckets is not None:
            self.buckets.zero_grad()                                              # gradients live in (and stay attached to) the buckets
        else:
            self.optim.zero_grad(set_to_none=True)
        b0 = mark() if mark else None
        total.backward(self.coefs / float(len(minibatch)))                        # :275 -> DTLoss.to_scalar: dot(coefs, losses) / count (utils.py:64-75)
        if self.buckets is not None:
            self.buckets.wait()                                                   # the all-reduces ran under the backward pass
        b1 = mark() if mark else None
        self.optim.step()
        b2 = mark() if mark else None
        return total, stamps, (b0, b1, b2)
