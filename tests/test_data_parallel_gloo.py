"""CPU, world_size 2, gloo: the gradient averaging of the data-parallel training step (SURVEY 8f-3,
BASELINE config 5) -- detect_to_track/data_parallel.py.  On the MI355X node the same code runs over
RCCL (backend "nccl"); the ops themselves have no cross-rank step (tests/test_sharding_gloo.py)."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(3)                                           # same weights on every rank
    return torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.ReLU(), torch.nn.Linear(40, 33), torch.nn.ReLU(),
                               torch.nn.Linear(33, 5))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT / "detect-to-track_amd" / "detect_to_track"))
    import data_parallel as dp                                     # (the package __init__ would load the HIP library: not needed here)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _model()
        extra = torch.nn.Parameter(torch.ones(7))                  # never used in the loss: gets no gradient
        params = list(m.parameters()) + [extra]
        gb = dp.GradientBuckets(params, bucket_mb=0.004)           # ~1000 floats per bucket: several buckets
        assert len(gb.buckets) >= 3
        seen = [p for b in gb.buckets for p in b.params]
        assert len(seen) == len(params) and {id(p) for p in seen} == {id(p) for p in params}   # a partition
        assert seen[0] is extra and seen[-1] is params[0]          # reverse registration order
        opt = torch.optim.SGD(params, lr=0.1)
        for step in range(3):
            torch.manual_seed(100 * step + rank)                   # different data per rank
            x, y = torch.randn(6, 12), torch.randn(6, 5)
            opt.zero_grad(set_to_none=True)
            loss = (m(x) - y).square().mean()
            loss.backward()
            local = [p.grad.clone() for p in m.parameters()]
            gb.wait()
            # expected: the mean over ranks of the local gradients
            for p, g in zip(m.parameters(), local):
                parts = [torch.empty_like(g) for _ in range(world)]
                dist.all_gather(parts, g)
                torch.testing.assert_close(p.grad, torch.stack(parts).mean(0), rtol=1e-6, atol=1e-7)
            assert extra.grad is not None and not extra.grad.any()  # no gradient on any rank: zeros (as DDP's unused parameters)
            opt.step()
        # after identical averaged updates the replicas still agree bit for bit
        for p in m.parameters():
            parts = [torch.empty_like(p.data) for _ in range(world)]
            dist.all_gather(parts, p.data)
            assert torch.equal(parts[0], parts[1])
        # gradient accumulation: two micro-batches, the first under no_sync(); the reduced value is the mean over
        # ranks of the SUM of both micro-batches' local gradients (the second one used to be dropped)
        gb.zero_grad()
        torch.manual_seed(900 + rank)
        xa, ya, xb, yb = torch.randn(6, 12), torch.randn(6, 5), torch.randn(6, 12), torch.randn(6, 5)
        la = [g.clone() for g in torch.autograd.grad((m(xa) - ya).square().mean(), list(m.parameters()))]
        lb = [g.clone() for g in torch.autograd.grad((m(xb) - yb).square().mean(), list(m.parameters()))]
        with gb.no_sync():
            (m(xa) - ya).square().mean().backward()
        (m(xb) - yb).square().mean().backward()
        gb.wait()
        for p, ga, gbb in zip(m.parameters(), la, lb):
            parts = [torch.empty_like(ga) for _ in range(world)]
            dist.all_gather(parts, ga + gbb)
            torch.testing.assert_close(p.grad, torch.stack(parts).mean(0), rtol=1e-5, atol=1e-7)
            assert p.grad.data_ptr() == gb.buckets[gb._where[p][0]].views[gb._where[p][1]].data_ptr()   # still the bucket view
        # a parameter used on rank 0 only (rank-asymmetric graph): both ranks must still issue the same collectives
        # in the same order (completion-order launches would hang or cross-match here)
        gb.zero_grad()
        torch.manual_seed(77 + rank)
        x = torch.randn(6, 12)
        loss = m(x).square().mean() + (extra.sum() * 3.0 if rank == 0 else 0.0)
        loss.backward()
        gb.wait()
        torch.testing.assert_close(extra.grad, torch.full((7,), 3.0 / world))      # mean of (3, 0)
        gb.remove()
        dist.barrier()
        q.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradient_buckets_two_ranks():
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


def test_single_process_is_a_no_op():
    sys.path.insert(0, str(ROOT / "detect-to-track_amd" / "detect_to_track"))
    import data_parallel as dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        m = _model()
        gb = dp.GradientBuckets(m.parameters())
        m(torch.randn(4, 12)).sum().backward()
        want = [p.grad.clone() for p in m.parameters()]
        gb.wait()
        assert all(torch.equal(p.grad, w) for p, w in zip(m.parameters(), want))
        with pytest.raises(ValueError):
            dp.GradientBuckets([torch.zeros(3)])
    finally:
        dist.destroy_process_group()
