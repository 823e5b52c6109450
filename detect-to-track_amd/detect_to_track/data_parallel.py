"""Data-parallel gradient averaging for the training step (BASELINE config 5).

The three custom ops have no cross-device step: a rank's frame pairs never meet another rank's
(SURVEY §8e).  What a multi-GPU trainer needs on top is the usual one collective -- averaging the
parameter gradients -- and on MI355X that is RCCL over xGMI (``torch.distributed`` backend "nccl").
``GradientBuckets`` does it the way that link topology wants it:

* gradients are packed into a few LARGE flat buckets (default 64 MB: xGMI is point-to-point, 7 links x
  ~153 GB/s per GPU, so a ring all-reduce is bound per link and pays its latency per call -- the model's
  37 M trainable parameters are 150 MB, i.e. three calls, not hundreds);
* buckets are filled in the order gradients become ready (reverse registration order) and a bucket's
  all-reduce is launched ASYNCHRONOUSLY the moment its last gradient arrives, from autograd's
  post-accumulate hooks, so that it runs under the rest of the backward pass (the backbone's early
  stages finish last and are the cheapest to wait for);
* ``wait()`` -- called once between ``backward()`` and ``optimizer.step()`` -- completes the handles,
  launches buckets that stayed incomplete (parameters that got no gradient this step count as
  zeros, on every rank alike), divides by the world size and hands the averages back to ``p.grad``.

Nothing here touches the data path of the ops; there is no CPU fallback to speak of either: on CPU
tensors with the gloo backend the same code runs unchanged, which is how tests/test_data_parallel_gloo.py
covers it (world_size 2).  It has NOT run on multi-GPU hardware yet.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class _Bucket:
    __slots__ = ("flat", "params", "offsets", "pending", "ready", "handle")

    def __init__(self, params: List[Tensor], dtype: torch.dtype, device: torch.device) -> None:
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.flat = torch.zeros(n, dtype=dtype, device=device)
        self.pending = len(params)
        self.ready = [False] * len(params)
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
        for b in self.buckets:
            for k, p in enumerate(b.params):
                self._where[p] = (b, k)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _launch(self, b: _Bucket) -> None:
        b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p: Tensor) -> None:
        b, k = self._where[p]
        if b.ready[k]:                                   # a second accumulation into the same parameter (gradient
            return                                       # accumulation over micro-batches): wait() picks up the final value
        n = p.numel()
        b.flat[b.offsets[k]: b.offsets[k] + n].copy_(p.grad.reshape(-1))
        b.ready[k] = True
        b.pending -= 1
        if b.pending == 0 and self.world > 1:
            self._launch(b)

    def wait(self) -> None:
        """Finish the step's all-reduces and leave the averaged gradients in ``p.grad``."""
        for b in self.buckets:
            if self.world > 1:
                if b.handle is None:                     # some parameter got no gradient: zeros, same on every rank
                    for k, p in enumerate(b.params):
                        if not b.ready[k]:
                            b.flat[b.offsets[k]: b.offsets[k] + p.numel()].zero_()
                    self._launch(b)
                b.handle.wait()
                b.flat.div_(self.world)
                for k, p in enumerate(b.params):
                    if p.grad is not None:
                        p.grad.copy_(b.flat[b.offsets[k]: b.offsets[k] + p.numel()].view_as(p.grad))
                    elif b.ready[k] is False and bool(b.flat[b.offsets[k]: b.offsets[k] + p.numel()].any()):
                        p.grad = b.flat[b.offsets[k]: b.offsets[k] + p.numel()].view_as(p).clone()
            b.handle = None
            b.pending = len(b.params)
            b.ready = [False] * len(b.params)

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
