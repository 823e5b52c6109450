"""CPU: pin the oracle against the golden fixtures produced by the reference's own kernels
(tests/golden/make_golden.py, run on MI355X with oracle/_ref).

Bit-exact: correlation forward and gradFM0 (both thread-owned FMA chains in the reference),
PSROIPool forward (same running-sum order), the written-cell mask, bin/cell bounds, channel map.
Tolerance: gradients the reference accumulates with atomicAdd (order undefined there): 1e-5 f32,
1e-12 f64; ROIPool forward: bit-exact as well (same running-sum order, unguarded divide).
"""
import numpy as np
import pytest

from conftest import golden_files, golden_ids, load_golden


def _tol(dtype):
    return dict(rtol=1e-5, atol=1e-5) if dtype == np.float32 else dict(rtol=1e-12, atol=1e-12)


def test_fixtures_present():
    assert len(golden_files("corr")) >= 20 and len(golden_files("roipool")) >= 20 and len(golden_files("psroipool")) >= 30


@pytest.mark.parametrize("path", golden_files("corr"), ids=golden_ids("corr"))
def test_correlation(path, oracle):
    g = load_golden(path)
    d, s = int(g["d"]), int(g["s"])
    np.testing.assert_array_equal(oracle.corr_fwd(g["fm0"], g["fm1"], d, s), g["out"])
    g0, g1 = oracle.corr_bwd(g["gout"], g["fm0"], g["fm1"], d, s)
    np.testing.assert_array_equal(g0, g["g0"])
    np.testing.assert_allclose(g1, g["g1"], **_tol(g["fm0"].dtype))
    H, W = g["fm0"].shape[2:]
    np.testing.assert_array_equal(oracle.corr_mask(H, W, d, s), g["mask"])


def _check_bounds(bounds, ref_bounds):
    empty = ref_bounds[..., 0] < 0
    ours_empty = (bounds[..., 1] <= bounds[..., 0]) | (bounds[..., 3] <= bounds[..., 2])
    np.testing.assert_array_equal(ours_empty, empty)
    np.testing.assert_array_equal(bounds[~empty], ref_bounds[~empty])


@pytest.mark.parametrize("path", golden_files("roipool"), ids=golden_ids("roipool"))
def test_roipool(path, oracle):
    g = load_golden(path)
    k = int(g["k"])
    _, H, W = g["fm"].shape
    np.testing.assert_array_equal(oracle.roipool_fwd(g["fm"], g["rois"], k), g["out"])     # NaNs compare equal
    np.testing.assert_allclose(oracle.roipool_bwd(g["gout"], g["rois"], H, W), g["gin"], **_tol(g["fm"].dtype))
    _check_bounds(oracle.roipool_bins(g["rois"], H, W, k), g["bounds"])


@pytest.mark.parametrize("path", [p for p in golden_files("psroipool") if "known_answer" not in p.stem],
                         ids=[i for i in golden_ids("psroipool") if "known_answer" not in i])
def test_psroipool(path, oracle):
    g = load_golden(path)
    nT, k = int(g["nT"]), int(g["k"])
    _, H, W = g["fm"].shape
    np.testing.assert_array_equal(oracle.psroipool_fwd(g["fm"], g["rois"], nT, k), g["out"])
    np.testing.assert_allclose(oracle.psroipool_bwd(g["gout"], g["rois"], H, W), g["gin"], **_tol(g["fm"].dtype))
    _check_bounds(oracle.roipool_bins(g["rois"], H, W, k, position_sensitive=True), g["bounds"])
    np.testing.assert_array_equal(oracle.psroipool_channels(nT, k), g["channels"])


def test_psroipool_known_answer(oracle):
    """reference tests/test_ps_roipool.py:33-44: constant map, RoI (3,3,.5,.5) -> all zeros."""
    g = load_golden([p for p in golden_files("psroipool") if "known_answer" in p.stem][0])
    fm = np.full((2 * 49, 10, 11), 10.0, dtype=np.float32)
    out = oracle.psroipool_fwd(fm, np.asarray([[3.0, 3.0, 0.5, 0.5]], np.float32), 2, 7)
    np.testing.assert_array_equal(out, g["out"])
    assert not out.any()


def test_oracle_headline_subsample(oracle):
    """The oracle at BASELINE.json's metric shape against the reference's values (seeded subsample)."""
    from conftest import GOLDEN
    path = GOLDEN / "corr_headline_subsample.npz"
    g = load_golden(path)
    B, C, H, W, d = (int(g[k]) for k in ("B", "C", "H", "W", "d"))
    rng = np.random.default_rng(int(g["seed"]))
    fm0, fm1 = rng.random((B, C, H, W), dtype=np.float32), rng.random((B, C, H, W), dtype=np.float32)
    gout = rng.random((B, H, W, 2 * d + 1, 2 * d + 1), dtype=np.float32)
    assert fm0.flat[12345] == g["fm0_probe"] and gout.flat[54321] == g["gout_probe"]
    out = oracle.corr_fwd(fm0, fm1, d, 1).ravel()
    np.testing.assert_array_equal(out[g["out_idx"]], g["out_val"])
    g0, g1 = oracle.corr_bwd(gout, fm0, fm1, d, 1)
    np.testing.assert_array_equal(g0.ravel()[g["g_idx"]], g["g0_val"])          # thread-owned order
    np.testing.assert_allclose(g1.ravel()[g["g_idx"]], g["g1_val"], rtol=1e-5, atol=1e-5)


# ---- the wide-accumulator yardsticks (oracle *_acc64): terms as the reference forms them, summed in double.
# Pinned here against the reference's own gradients of the fixtures: every f32 fixture gradient (atomicAdd order
# on the GPU that made it) lies within 1e-5 of sum|terms| of the yardstick, and the yardstick's geometry / terms
# are the pinned oracle's (same bin functions), so it may stand in for "the reference's value" at shapes where an
# f32 running sum of any order is noisier than the contract.
@pytest.mark.parametrize("path", [p for p in golden_files("corr")], ids=golden_ids("corr"))
def test_acc64_yardstick_correlation(path, oracle):
    g = load_golden(path)
    (w0, w1), (m0, m1) = oracle.corr_bwd_acc64(g["gout"], g["fm0"], g["fm1"], int(g["d"]), int(g["s"]))
    rel = 1e-5 if g["fm0"].dtype == np.float32 else 1e-12
    oracle.assert_within_contract(g["g0"], w0, m0, rel, "gradFM0")
    oracle.assert_within_contract(g["g1"], w1, m1, rel, "gradFM1")


@pytest.mark.parametrize("path", golden_files("roipool"), ids=golden_ids("roipool"))
def test_acc64_yardstick_roipool(path, oracle):
    g = load_golden(path)
    _, H, W = g["fm"].shape
    want, mag = oracle.roipool_bwd_acc64(g["gout"], g["rois"], H, W)
    oracle.assert_within_contract(g["gin"], want, mag, 1e-5 if g["fm"].dtype == np.float32 else 1e-12, "ROIPool gradient")


@pytest.mark.parametrize("path", [p for p in golden_files("psroipool") if "known_answer" not in p.stem],
                         ids=[i for i in golden_ids("psroipool") if "known_answer" not in i])
def test_acc64_yardstick_psroipool(path, oracle):
    g = load_golden(path)
    _, H, W = g["fm"].shape
    want, mag = oracle.psroipool_bwd_acc64(g["gout"], g["rois"], H, W)
    oracle.assert_within_contract(g["gin"], want, mag, 1e-5 if g["fm"].dtype == np.float32 else 1e-12, "PSROIPool gradient")


def test_fixtures_record_the_reference_sources_they_were_made_from():
    """Every fixture carries the sha256 of the reference files (and of oracle/ref_build/torch_compat.h, the one adaptation)
    that oracle/_ref was compiled from when it was generated.  All fixtures agree; where /root/reference is mounted (the
    build container) the hashes are re-derived from the files as they lie there today."""
    import hashlib
    from pathlib import Path
    from conftest import GOLDEN
    files = sorted(GOLDEN.glob("*.npz"))
    stamps = set()
    for f in files:
        g = load_golden(f)
        assert "ref_sources_sha256" in g, f"{f.name} has no source stamp: regenerate with tests/golden/make_golden.py"
        stamps.add(str(g["ref_sources_sha256"]))
    assert len(stamps) == 1, "fixtures were generated from different source states"
    listed = dict(reversed(line.split(None, 1)) for line in stamps.pop().strip().splitlines())
    assert len(listed) == 9 and "oracle/ref_build/torch_compat.h" in listed and "ps_roipool/ps_roipool_cuda.cu" in listed
    ref = Path("/root/reference/detect_to_track/models")
    root = Path(__file__).resolve().parents[1]
    for name, digest in listed.items():
        name = name.strip()
        path = root / name if name.startswith("oracle/") else ref / name
        if path.exists():                                             # the reference is absent on the GPU box
            assert hashlib.sha256(path.read_bytes()).hexdigest() == digest, f"{name} changed since the fixtures were made"
