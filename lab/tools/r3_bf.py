"""Round-3 development check: the bf16x3 correlation backward (impl 4) against the type-generic kernel, determinism, timing."""
import sys
sys.path.insert(0, "detect-to-track_amd")
import torch
from detect_to_track.models import _ext

dev = "cuda:0"
ok = True
for shape in [(3, 20, 38, 75), (2, 300, 21, 44), (1, 256, 38, 63), (2, 100, 40, 24), (1, 64, 17, 130), (8, 256, 38, 63)]:
    B, C, H, W = shape
    g = torch.Generator().manual_seed(1)
    fm0, fm1 = torch.randn(B, C, H, W, generator=g).to(dev), torch.randn(B, C, H, W, generator=g).to(dev)
    gout = torch.randn(B, H, W, 17, 17, generator=g).to(dev)
    r0, r1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, 1)
    scale = float(r0.abs().max())
    for impl in (0, 4):
        g0, g1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, impl)
        e0, e1 = float((g0 - r0).abs().max()) / scale, float((g1 - r1).abs().max()) / scale
        h0, h1 = _ext.pointwise_correlation_backward(gout, fm0, fm1, 8, 1, impl)
        det = torch.equal(g0, h0) and torch.equal(g1, h1)
        good = max(e0, e1) < 3e-6 and det
        ok &= good
        print(shape, "impl", impl, "max|err|/max|g| =", f"{e0:.2e} {e1:.2e}", "deterministic" if det else "NOT DETERMINISTIC", "OK" if good else "FAIL", flush=True)
B, C, H, W = 8, 256, 38, 63
sets = []
for i in range(6):
    g = torch.Generator().manual_seed(i)
    sets.append((torch.rand(B, H, W, 17, 17, generator=g).to(dev), torch.rand(B, C, H, W, generator=g).to(dev),
                 torch.rand(B, C, H, W, generator=g).to(dev)))
res = {0: [], 4: []}
for rnd in range(5):
    for impl in (0, 4):
        for k in range(6):
            _ext.pointwise_correlation_backward(*sets[k], 8, 1, impl)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(60):
            _ext.pointwise_correlation_backward(*sets[k % 6], 8, 1, impl)
        b.record(); torch.cuda.synchronize()
        res[impl].append(a.elapsed_time(b) / 60 * 1e3)
for impl in (0, 4):
    v = sorted(res[impl])
    print("impl", impl, "us per backward (incl. allocation of outputs): median", round(v[len(v) // 2], 1), "min", round(v[0], 1))
print("ALL OK" if ok else "FAILURES")
