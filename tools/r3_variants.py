"""Round-3 development check: every backward variant (impl codes 0, 3..6) against the type-generic kernel, then timing."""
import sys, time
sys.path.insert(0, "detect-to-track_amd")
import torch
from detect_to_track.models import _ext

dev = "cuda:0"
for shape in [(3, 20, 38, 75), (2, 300, 21, 44), (8, 256, 38, 63)]:
    B, C, H, W = shape
    g = torch.Generator().manual_seed(1)
    fm0, fm1 = torch.rand(B, C, H, W, generator=g).to(dev), torch.rand(B, C, H, W, generator=g).to(dev)
    gout = torch.randn(B, H, W, 17, 17, generator=g).to(dev)
    r0, r1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, 1)
    scale = float(r0.abs().max())
    for impl in (0, 3, 4):
        g0, g1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, impl)
        e0, e1 = float((g0 - r0).abs().max()) / scale, float((g1 - r1).abs().max()) / scale
        h0, h1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, impl)
        det = torch.equal(g0, h0) and torch.equal(g1, h1)
        print(shape, "impl", impl, "max|err|/max|g| =", f"{e0:.2e} {e1:.2e}", "deterministic" if det else "NOT DETERMINISTIC",
              "OK" if max(e0, e1) < 2e-6 else "FAIL")
# timing at the metric shape, interleaved rounds in one process
B, C, H, W = 8, 256, 38, 63
sets = []
for i in range(6):
    g = torch.Generator().manual_seed(i)
    sets.append((torch.rand(B, H, W, 17, 17, generator=g).to(dev), torch.rand(B, C, H, W, generator=g).to(dev),
                 torch.rand(B, C, H, W, generator=g).to(dev)))
impls = [int(x) for x in sys.argv[1:]] or [3, 0, 4, 5, 6]
res = {i: [] for i in impls}
for rnd in range(5):
    for impl in impls:
        for k in range(6):
            _ext.pointwise_correlation_backward(*sets[k], 8, 1, impl)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(60):
            _ext.pointwise_correlation_backward(*sets[k % 6], 8, 1, impl)
        b.record(); torch.cuda.synchronize()
        res[impl].append(a.elapsed_time(b) / 60 * 1e3)
for impl in impls:
    v = sorted(res[impl])
    print("impl", impl, "us per backward (incl. allocation of outputs): median", round(v[len(v) // 2], 1), "min", round(v[0], 1))
