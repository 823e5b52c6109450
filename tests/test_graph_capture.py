"""The C-ABI entry points are asynchronous, allocate nothing on the device themselves, keep no state
and never synchronise (include/d2t_ops.h): so a whole forward+backward of the three
ops must be capturable into one HIP graph and replayable on new data.  torch's graph capture fails
on any synchronising or illegal call inside the captured region, which makes it a sharp test of
those claims; the replayed results are then compared with eager launches on the same inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _all_ops(t):
    from detect_to_track.models import _ext
    out = _ext.pointwise_correlation_forward(t["fm0"], t["fm1"], 8, 1)
    g0, g1 = _ext.pointwise_correlation_backward(t["gout"], t["fm0"], t["fm1"], 8, 1)
    rp = _ext.roipool_forward(t["fm"], t["rois"], 7)
    rb = _ext.roipool_backward(t["rgo"], t["rois"], t["fm"].shape[1], t["fm"].shape[2])
    ps = _ext.ps_roipool_forward(t["pfm"], t["rois"], 3, 7)
    pb = _ext.ps_roipool_backward(t["pgo"], t["rois"], t["pfm"].shape[1], t["pfm"].shape[2])
    return out, g0, g1, rp, rb, ps, pb


def _inputs(seed, R=40):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g).to(DEV)
    rois = torch.cat([r(R, 2), 0.05 + 0.55 * r(R, 2)], 1).contiguous()
    return dict(fm0=r(2, 48, 21, 33), fm1=r(2, 48, 21, 33), gout=r(2, 21, 33, 17, 17),
                fm=r(96, 21, 33), rgo=r(R, 96, 7, 7), pfm=r(3 * 49, 21, 33), pgo=r(R, 3, 7, 7), rois=rois)


def test_forward_backward_of_all_ops_in_one_graph():
    static = _inputs(1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # warm-up outside the capture (library load, allocator)
        _all_ops(static)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = _all_ops(static)
    for seed in (2, 3):
        fresh = _inputs(seed)
        for k, v in fresh.items():
            static[k].copy_(v)
        graph.replay()
        torch.cuda.synchronize()
        eager = _all_ops(fresh)
        for name, c, e in zip(("out", "g0", "g1", "roipool", "roipool_bwd", "psroipool", "psroipool_bwd"), captured, eager):
            assert torch.equal(torch.nan_to_num(c, nan=-7.0), torch.nan_to_num(e, nan=-7.0)), name   # deterministic kernels: bitwise


def _all_ops_second_tier(t):
    """The same outside the tuned envelope (d_max 7, k 6): the second kernel tier -- pre-passes into the workspace, a counter zeroed by a
    memset node, device-side gates -- must be capturable too."""
    from detect_to_track.models import _ext
    out = _ext.pointwise_correlation_forward(t["fm0"], t["fm1"], 7, 1)
    g0, g1 = _ext.pointwise_correlation_backward(t["gout"], t["fm0"], t["fm1"], 7, 1)
    rp = _ext.roipool_forward(t["fm"], t["rois"], 6)
    rb = _ext.roipool_backward(t["rgo"], t["rois"], t["fm"].shape[1], t["fm"].shape[2])
    ps = _ext.ps_roipool_forward(t["pfm"], t["rois"], 3, 6)
    pb = _ext.ps_roipool_backward(t["pgo"], t["rois"], t["pfm"].shape[1], t["pfm"].shape[2])
    return out, g0, g1, rp, rb, ps, pb


def _inputs_second_tier(seed, R=40):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g).to(DEV)
    rois = torch.cat([r(R, 2), 0.05 + 0.55 * r(R, 2)], 1).contiguous()
    return dict(fm0=r(2, 48, 21, 33), fm1=r(2, 48, 21, 33), gout=r(2, 21, 33, 15, 15),
                fm=r(96, 21, 33), rgo=r(R, 96, 6, 6), pfm=r(3 * 36, 21, 33), pgo=r(R, 3, 6, 6), rois=rois)


def test_second_tier_ops_in_one_graph():
    import warnings
    warnings.simplefilter("ignore")                      # the wrappers' once-per-op envelope warning
    static = _inputs_second_tier(1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _all_ops_second_tier(static)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = _all_ops_second_tier(static)
    for seed in (2, 3):
        fresh = _inputs_second_tier(seed)
        for k, v in fresh.items():
            static[k].copy_(v)
        graph.replay()
        torch.cuda.synchronize()
        eager = _all_ops_second_tier(fresh)
        for name, c, e in zip(("out", "g0", "g1", "roipool", "roipool_bwd", "psroipool", "psroipool_bwd"), captured, eager):
            assert torch.equal(torch.nan_to_num(c, nan=-7.0), torch.nan_to_num(e, nan=-7.0)), name
