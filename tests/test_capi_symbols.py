"""CPU: the C-ABI library loads, exports every symbol include/d2t_ops.h declares, and rejects bad
arguments with the documented codes BEFORE touching a device (no compute without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "d2t_ops.h").read_text()


def declared_functions():
    names = re.findall(r"^\s*(?:int|size_t|const char\*)\s+(d2t_\w+)\s*\(", HEADER, flags=re.M)
    assert len(names) >= 26
    return sorted(set(names))


@pytest.fixture(scope="module")
def native():
    from detect_to_track.models import _native
    return _native


@pytest.mark.parametrize("name", declared_functions())
def test_symbol_exported(name, native):
    assert hasattr(native.lib, name), f"{name} declared in include/d2t_ops.h but not exported"
    assert name in native.SYMBOLS, f"{name} has no ctypes prototype in _native.py"


def test_version_and_error_strings(native):
    assert native.version() >= 107                    # round 6: exact ROIPool forward, no workspace for the one-launch PSROIPool forward
    assert native.error_string(0) == b"ok"
    for code in (-1, -2, -3):
        assert native.error_string(code) and native.error_string(code) != b"ok"


def test_argument_validation_without_device(native):
    L = native.lib
    EINVAL, ETOOBIG, EWS = -1, -2, -3
    # null pointers with non-empty extents
    assert L.d2t_corr_fwd_f32(None, None, None, 1, 2, 5, 5, 2, 1, None, 0, 0, None) == EINVAL
    # stride < 1, negative extent
    assert L.d2t_corr_fwd_f32(8, 8, 8, 1, 2, 5, 5, 2, 0, None, 0, 0, None) == EINVAL
    assert L.d2t_corr_fwd_f64(8, 8, 8, -1, 2, 5, 5, 2, 1, None, 0, 0, None) == EINVAL
    # output element count beyond int32 (reference offsets are int: pointwise_correlation_cuda.cu:44-50)
    assert L.d2t_corr_fwd_f32(8, 8, 8, 64, 2, 512, 512, 8, 1, None, 0, 0, None) == ETOOBIG
    assert L.d2t_roipool_fwd_f32(8, 8, 8, 300000, 1024, 38, 63, 7, None, 0, 0, None) == ETOOBIG
    # k < 1
    assert L.d2t_roipool_fwd_f32(8, 8, 8, 1, 1, 5, 5, 0, None, 0, 0, None) == EINVAL
    assert L.d2t_psroipool_fwd_f64(8, 8, 8, 1, 1, 5, 5, 0, None, 0, 0, None) == EINVAL
    # backward needs its bin-table workspace
    assert L.d2t_roipool_bwd_f64(8, 8, 8, 3, 2, 10, 10, 5, None, 0, 1, None) == EWS
    assert L.d2t_psroipool_bwd_f64(8, 8, 8, 3, 2, 10, 10, 5, None, 0, 1, None) == EWS
    need = L.d2t_roipool_bwd_workspace_bytes(3, 2, 10, 10, 5, 8)
    assert need >= 3 * 5 * 5 * 4 * 4
    assert L.d2t_psroipool_bwd_workspace_bytes(3, 2, 10, 10, 5, 8) >= 3 * 5 * 5 * 4 * 4
    # a workspace pointer that is not 16-byte aligned is refused before anything touches the device
    assert L.d2t_roipool_bwd_f32(16, 16, 16, 3, 2, 10, 10, 5, 0x1008, 1 << 20, 0, None) == EINVAL
    assert L.d2t_corr_bwd_f32(16, 16, 16, 16, 16, 1, 2, 9, 9, 3, 1, 0x1004, 1 << 20, 0, None) == EINVAL
    # the tuned-only selector refuses shapes the tuned kernels do not take instead of silently falling back
    assert L.d2t_corr_fwd_f32(8, 8, 8, 1, 2, 10, 10, 3, 1, None, 0, 2, None) == EINVAL
    assert L.d2t_corr_fwd_f64(8, 8, 8, 1, 2, 38, 63, 8, 1, None, 0, 2, None) == EINVAL
    # introspection entry points validate too
    assert L.d2t_corr_mask(None, 5, 5, 2, 1, None) == EINVAL
    assert L.d2t_psroipool_channels(None, 2, 3, None) == EINVAL
    assert L.d2t_roipool_bins_f32(None, None, 2, 5, 5, 3, None) == EINVAL


def test_workspace_queries_cover_the_second_kernel_tier(native):
    """Outside the tuned envelope the *_workspace_bytes queries ask for the scratch of the kernels the default dispatch takes there
    (include/d2t_ops.h: zero-padded map copies + gradOut by displaced pixel for the correlation backward; bin lists + gradOut / n by
    (bin, channel) for the pooling backward) -- host arithmetic only, no device needed.  Inside the envelope nothing changes."""
    L = native.lib
    B, C, H, W = 8, 256, 38, 63
    cells7 = 15 * 15
    tiled = L.d2t_corr_bwd_workspace_bytes(B, C, H, W, 7, 1, 4)
    assert tiled >= B * H * W * cells7 * 4 + 2 * B * C * H * (W + 14) * 4        # gradOut re-indexed + both maps with padded rows
    assert L.d2t_corr_bwd_workspace_bytes(B, C, H, W, 20, 1, 4) >= B * H * W * 41 * 41 * 4   # d_max > 14: the blocked kernels (no map copies)
    assert L.d2t_corr_bwd_workspace_bytes(B, C, H, W, 20, 1, 4) < B * H * W * 41 * 41 * 4 + 2 * B * C * H * W * 4
    assert L.d2t_corr_bwd_workspace_bytes(B, C, H, W, 7, 1, 8) >= B * H * W * cells7 * 8     # f64: blocked
    R, Cp, k = 300, 1024, 6
    lists = L.d2t_roipool_bwd_workspace_bytes(R, Cp, H, W, k, 4)
    assert lists >= R * Cp * k * k * 4 + R * (H + 2 * k) * 4 + R * W * 4 + R * k * k * 16    # gradOut / n by (bin, channel), row lists, column masks, bins
    assert L.d2t_roipool_bwd_workspace_bytes(R, Cp, H, W, k, 8) >= R * Cp * k * k * 8
    ps = L.d2t_psroipool_bwd_workspace_bytes(R, 21, H, W, k, 4)
    assert ps >= R * k * (H + 2 * k) * 16 + k * k * H * 8                       # (RoI, area, column range) per (cell, map row) + list heads
    # k = 7, f32: the tuned kernels' own (much smaller) scratch -- the lists are not asked for
    assert L.d2t_roipool_bwd_workspace_bytes(R, Cp, H, W, 7, 4) < R * Cp * 49 * 4


def test_header_cites_the_reference_binding():
    # every replaced pybind function is named with its file:line
    for token in ("pointwise_correlation.cpp:23-33", "pointwise_correlation.cpp:36-48", "roipool.cpp:22-32",
                  "roipool.cpp:35-45", "ps_roipool.cpp:23-34", "ps_roipool.cpp:37-47"):
        assert token in HEADER


def test_workspace_queries_round6(native):
    """ABI 1.07: the PSROIPool forward needs NO workspace where it is the one-launch bin-row kernel (a target is one share of <= 1,024 RoIs: the
    model's class / regression heads, BASELINE config 3) -- ADVICE r5: one predicate for the query and the launcher --, and still asks for the RoI
    order / the transposed planes elsewhere; the ROIPool forward never needs one."""
    L = native.lib
    assert L.d2t_psroipool_fwd_workspace_bytes(300, 21, 38, 63, 7, 4) == 0       # config 3
    assert L.d2t_psroipool_fwd_workspace_bytes(300, 31, 38, 75, 7, 4) == 0
    assert L.d2t_psroipool_fwd_workspace_bytes(3000, 4, 38, 75, 7, 4) >= 3000 * 4  # several shares of the RoI list: the order comes from a pre-pass
    assert L.d2t_psroipool_fwd_workspace_bytes(3, 3, 20, 30, 7, 4) == 0           # small: thread-per-output kernel
    for R in (8, 300):
        assert L.d2t_roipool_fwd_workspace_bytes(R, 1891, 38, 75, 7, 4) == 0
