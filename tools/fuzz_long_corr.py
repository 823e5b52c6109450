"""dev helper: long randomised sweep of the correlation kernels, tuned vs generic (forward bit-exact, gradients
against the f32 rounding of the sum of term magnitudes)."""
import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "detect-to-track_amd"))
from detect_to_track.models import _ext, _native
GENERIC, TUNED = 1, 2
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for it in range(N):
    B = int(rng.integers(1, 10)); C = int(rng.integers(1, 300)); H = int(rng.integers(1, 50)); W = int(rng.integers(20, 100))
    if len(sys.argv) > 3 and sys.argv[3] == "wide":                 # wide / tall maps, few channels: grids of other proportions than the model's
        B = int(rng.integers(1, 4)); C = int(rng.integers(1, 40)); H = int(rng.integers(1, 130)); W = int(rng.integers(20, 420))
    g = torch.Generator(device="cpu").manual_seed(it * 7919 + 13)
    fm0 = (torch.rand(B, C, H, W, generator=g) - 0.5).cuda(); fm1 = (torch.rand(B, C, H, W, generator=g) - 0.5).cuda()
    go = torch.randn(B, H, W, 17, 17, generator=g).cuda()
    try:
        a, b = _ext.pointwise_correlation_forward(fm0, fm1, 8, 1, TUNED), _ext.pointwise_correlation_forward(fm0, fm1, 8, 1, GENERIC)
        assert torch.equal(a, b), f"forward max |delta| {float((a - b).abs().max())}"
        t0, t1 = _ext.pointwise_correlation_backward(go, fm0, fm1, 8, 1, TUNED if H >= 17 else 0)   # the tuned backward needs 17 map rows (ABI 1.06); below: default dispatch
        g0, g1 = _ext.pointwise_correlation_backward(go, fm0, fm1, 8, 1, GENERIC)
        m0, m1 = _ext.pointwise_correlation_backward(go.abs(), fm0.abs(), fm1.abs(), 8, 1, GENERIC)
        assert bool(((t0 - g0).abs() <= 4e-6 * m0 + 1e-7).all()) and bool(((t1 - g1).abs() <= 4e-6 * m1 + 1e-7).all())
        if H >= 17:                                                  # lab build: the bf16x3 backward and the two 8-wave kernels demanded; product: the tuned backward demanded -- same bar
            for impl, name in (((4, "bf16x3"), (6, "strips 8 pixels wide"), (7, "strips 4 pixels wide")) if _native.IS_LAB_BUILD else ((2, "tuned kernels demanded"),)):
                s0, s1 = _ext.pointwise_correlation_backward(go, fm0, fm1, 8, 1, impl)
                assert bool(((s0 - g0).abs() <= 4e-6 * m0 + 1e-7).all()) and bool(((s1 - g1).abs() <= 4e-6 * m1 + 1e-7).all()), name
        fa = _ext.pointwise_correlation_forward(fm0, fm1, 8, 1, 5)   # D2T_IMPL_FAST: may split channels (same terms, other association)
        mf = _ext.pointwise_correlation_forward(fm0.abs(), fm1.abs(), 8, 1, GENERIC)
        assert bool(((fa - b).abs() <= 4e-6 * mf + 1e-7).all()), "fast forward"
    except Exception as e:
        bad += 1; print("CORR FAIL", (B, C, H, W), str(e)[:200], flush=True)
    if it % 20 == 19: print(f"{it + 1} cases, {bad} failures", flush=True)
print("done", N, "cases,", bad, "failures")
