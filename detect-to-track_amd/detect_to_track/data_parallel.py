"""Data-parallel gradient averaging for the training step (BASELINE config 5).

The three custom ops have no cross-device step: a rank's frame pairs never meet another rank's
(SURVEY §8e).  What a multi-GPU trainer needs on top is the usual one collective -- averaging the
parameter gradients -- and on MI355X that is RCCL over xGMI (``torch.distributed`` backend "nccl").
``GradientBuckets`` does it the way that link topology wants it:

* gradients are packed into a few LARGE flat buckets (default 64 MB: xGMI is point-to-point, 7 links x
  ~153 GB/s per GPU, so a ring all-reduce is bound per link and pays its latency per call -- the model's
  37 M trainable parameters are 150 MB, i.e. three calls, not hundreds);
* ``p.grad`` IS a view of its bucket (gradient-as-bucket-view): autograd accumulates straight into the
  flat buffer and the averaged values are read from it by the optimizer -- no copy in, no copy out
  (2 x 150 MB per step otherwise);
* buckets are filled in the order gradients become ready (reverse registration order) and all-reduced
  ASYNCHRONOUSLY from autograd's post-accumulate hooks, under the rest of the backward pass (the
  backbone's early stages finish last and are the cheapest to wait for) -- always in BUCKET ORDER: a bucket
  whose last gradient has arrived waits for its predecessors, so that every rank issues the same sequence
  of collectives even when a parameter gets a gradient on one rank and none on another;
* ``wait()`` -- called once between the last ``backward()`` and ``optimizer.step()`` -- launches the
  buckets that stayed incomplete (a parameter without a gradient this step contributes the zeros
  ``zero_grad()`` left in its view), completes the handles and divides by the world size;
* gradient accumulation: run every micro-batch but the last under ``with buckets.no_sync():`` -- the hooks
  then only let autograd accumulate; the last ``backward()`` reduces the sums.

Use ``buckets.zero_grad()`` instead of ``optimizer.zero_grad()`` (which, with ``set_to_none=True``, would
detach the gradients from the buckets; a gradient found detached is copied back and re-attached, so that
is a slow path, not an error).  Like DistributedDataParallel with ``find_unused_parameters``, a parameter
that got no gradient on any rank ends the step with a zero gradient rather than ``None``.

Nothing here touches the data path of the ops; on CPU tensors with the gloo backend the same code runs
unchanged, which is how tests/test_data_parallel_gloo.py covers it (world_size 2).  It has NOT run on
multi-GPU hardware yet.
"""
import contextlib
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class _Bucket:
    __slots__ = ("flat", "params", "offsets", "views", "pending", "ready", "seen", "handle")

    def __init__(self, params: List[Tensor], dtype: torch.dtype, device: torch.device) -> None:
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.flat = torch.zeros(n, dtype=dtype, device=device)
        self.views = [self.flat[o: o + p.numel()].view_as(p) for o, p in zip(self.offsets, params)]
        self.pending = len(params)
        self.ready = [False] * len(params)
        self.seen = [False] * len(params)                # got a gradient in this step (any micro-batch)
        self.handle = None


class GradientBuckets:
    """Average the gradients of ``params`` over the process group with bucketed, overlapped all-reduces.

    Args:
        params: the trainable parameters (one dtype, one device).
        bucket_mb: bucket capacity in MiB of gradient.
        group: process group (default: the world).
    """

    def __init__(self, params: Iterable[Tensor], bucket_mb: float = 64.0, group: Optional[dist.ProcessGroup] = None) -> None:
        plist = [p for p in params if p.requires_grad]
        if not plist:
            raise ValueError("no trainable parameters")
        if len({(p.dtype, p.device) for p in plist}) != 1:
            raise ValueError("parameters must share one dtype and one device")
        self.group = group
        self.world = dist.get_world_size(group)
        cap = max(1, int(bucket_mb * (1 << 20)) // plist[0].element_size())
        # gradients arrive roughly in reverse registration order: fill the buckets in that order
        self.buckets: List[_Bucket] = []
        cur, cur_n = [], 0
        for p in reversed(plist):
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(_Bucket(cur, p.dtype, p.device))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        self.buckets.append(_Bucket(cur, plist[0].dtype, plist[0].device))
        self._where = {}
        self._hooks = []
        self._sync = True
        self._next = 0                                   # first bucket whose all-reduce has not been launched this step
        for i, b in enumerate(self.buckets):
            for k, p in enumerate(b.params):
                self._where[p] = (i, k)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.zero_grad()

    # ------------------------------------------------------------------ gradients live in the buckets
    def zero_grad(self) -> None:
        """Zero every gradient and (re-)attach ``p.grad`` to its bucket view."""
        for b in self.buckets:
            b.flat.zero_()
            for p, v in zip(b.params, b.views):
                p.grad = v

    @contextlib.contextmanager
    def no_sync(self):
        """Micro-batches whose gradients are only accumulated (every ``backward()`` but the step's last)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    # ------------------------------------------------------------------ collectives, in bucket order
    def _launch_ready(self, force: bool = False) -> None:
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b.pending and not force:
                return
            if self.world > 1:
                b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._next += 1

    def _on_grad(self, p: Tensor) -> None:
        i, k = self._where[p]
        b = self.buckets[i]
        if p.grad is not b.views[k] and p.grad.data_ptr() != b.views[k].data_ptr():
            # detached by optimizer.zero_grad(set_to_none=True) or by the caller: autograd has just created this
            # tensor with the step's first gradient of p; it moves into the bucket (whose content is last step's)
            # and later micro-batches accumulate there
            b.views[k].copy_(p.grad)
            p.grad = b.views[k]
        b.seen[k] = True
        if not self._sync or b.ready[k]:
            return
        b.ready[k] = True
        b.pending -= 1
        if b.pending == 0:
            self._launch_ready()

    def wait(self) -> None:
        """Finish the step's all-reduces; afterwards every ``p.grad`` holds the average over the ranks."""
        for b in self.buckets[self._next:]:              # not launched: some parameter got no gradient on this rank --
            for k, p in enumerate(b.params):             # it contributes zeros (left by zero_grad(), or written here if
                if not b.seen[k] and p.grad is None:     # the caller detached the gradient with set_to_none)
                    b.views[k].zero_()
        self._launch_ready(force=True)
        for b in self.buckets:
            if b.handle is not None:
                b.handle.wait()
                b.flat.div_(self.world)
            for p, v in zip(b.params, b.views):          # a parameter unused on THIS rank may have a gradient from others
                if p.grad is None:
                    p.grad = v
            b.handle = None
            b.pending = len(b.params)
            b.ready = [False] * len(b.params)
            b.seen = [False] * len(b.params)
        self._next = 0

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
