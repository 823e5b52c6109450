"""ctypes binding of libd2t_ops.so (C ABI declared in /include/d2t_ops.h).

The library is loaded when ``detect_to_track.models`` is imported -- the same moment the
reference JIT-builds its CUDA extension (reference pointwise_correlation.py:13-22) -- and the
import FAILS LOUDLY if it is missing: there is no CPU or PyTorch fallback for these ops.
"""
import ctypes
import os
from ctypes import c_char_p, c_int, c_size_t, c_void_p
from pathlib import Path

# torch MUST be imported before libd2t_ops.so is mapped.  The PyTorch-ROCm wheel bundles its own
# HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7); once it is loaded the dynamic
# linker resolves our DT_NEEDED libamdhip64.so.7 to that same object, so kernels, streams and
# device pointers live in ONE runtime.  Loaded the other way round, the system copy under
# /opt/rocm would be mapped first and torch would add a second runtime next to it: launches from
# this library then fail with hipErrorNoDevice (seen on the MI355X box).
import torch  # noqa: F401

_HERE = Path(__file__).resolve().parent
_LIB_ENV = "D2T_OPS_LIBRARY"
_DEFAULT = _HERE.parent.parent / "lib" / "libd2t_ops.so"

IMPL_AUTO, IMPL_GENERIC, IMPL_MFMA, IMPL_FAST = 0, 1, 2, 5       # include/d2t_ops.h
# selectors of the LAB build only (lab/csrc/d2t_lab_selectors.h; `make -C csrc lab`, D2T_OPS_LIBRARY=.../lib_lab/libd2t_ops.so):
# the product library rejects them with D2T_EINVAL
LAB_IMPL_STRIP16, LAB_IMPL_BF16X3, LAB_IMPL_WIDE8, LAB_IMPL_STRIP4 = 3, 4, 6, 7


def _locate() -> Path:
    override = os.environ.get(_LIB_ENV)
    path = Path(override) if override else _DEFAULT
    if not path.is_file():
        raise ImportError(
            f"libd2t_ops.so not found at {path}. Build it with "
            f"`make -C {_HERE.parent.parent / 'csrc'}` (or `python -c 'import __graft_entry__ as g; g.build()'` "
            f"at the repository root), or point {_LIB_ENV} at it. The detect_to_track ops have no "
            "fallback implementation."
        )
    return path


LIBRARY_PATH = _locate()
lib = ctypes.CDLL(str(LIBRARY_PATH))

_P, _I, _Z = c_void_p, c_int, c_size_t


def _proto(name, restype, argtypes):
    fn = getattr(lib, name)          # AttributeError here = header and library disagree
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


version = _proto("d2t_version", _I, [])
error_string = _proto("d2t_error_string", c_char_p, [_I])

SYMBOLS = ["d2t_version", "d2t_error_string"]
for _t in ("f32", "f64"):
    # correlation: (fm0, fm1, out, B,C,H,W,d,stride, ws, ws_bytes, impl, stream)
    _proto(f"d2t_corr_fwd_{_t}", _I, [_P, _P, _P] + [_I] * 6 + [_P, _Z, _I, _P])
    # (gout, fm0, fm1, gfm0, gfm1, B,C,H,W,d,stride, ws, ws_bytes, impl, stream)
    _proto(f"d2t_corr_bwd_{_t}", _I, [_P] * 5 + [_I] * 6 + [_P, _Z, _I, _P])
    for _op in ("roipool", "psroipool"):
        # (fm|gout, rois, out|gin, R, C|nT, H, W, k, ws, ws_bytes, impl, stream)
        _proto(f"d2t_{_op}_fwd_{_t}", _I, [_P, _P, _P] + [_I] * 5 + [_P, _Z, _I, _P])
        _proto(f"d2t_{_op}_bwd_{_t}", _I, [_P, _P, _P] + [_I] * 5 + [_P, _Z, _I, _P])
        # (rois, bounds, R, H, W, k, stream)
        _proto(f"d2t_{_op}_bins_{_t}", _I, [_P, _P] + [_I] * 4 + [_P])
        SYMBOLS += [f"d2t_{_op}_fwd_{_t}", f"d2t_{_op}_bwd_{_t}", f"d2t_{_op}_bins_{_t}"]
    SYMBOLS += [f"d2t_corr_fwd_{_t}", f"d2t_corr_bwd_{_t}"]
for _n in ("corr_fwd", "corr_bwd"):
    _proto(f"d2t_{_n}_workspace_bytes", _Z, [_I] * 7)
    SYMBOLS.append(f"d2t_{_n}_workspace_bytes")
for _n in ("roipool_fwd", "roipool_bwd", "psroipool_fwd", "psroipool_bwd"):
    _proto(f"d2t_{_n}_workspace_bytes", _Z, [_I] * 6)
    SYMBOLS.append(f"d2t_{_n}_workspace_bytes")
# (n, fm0[], fm1[], out[], C[], B,H,W,d,stride, layout, batch_stride, ws, ws_bytes, impl, stream)
_proto("d2t_corr_fwd_levels_f32", _I, [_I, _P, _P, _P, _P] + [_I] * 5 + [_I, ctypes.c_longlong, _P, _Z, _I, _P])
# (n, gout[], fm0[], fm1[], gfm0[], gfm1[], C[], B,H,W,d,stride, layout, batch_stride, ws, ws_bytes, impl, stream)
_proto("d2t_corr_bwd_levels_f32", _I, [_I, _P, _P, _P, _P, _P, _P] + [_I] * 5 + [_I, ctypes.c_longlong, _P, _Z, _I, _P])
# (n, C[], B,H,W,d,stride)
_proto("d2t_corr_fwd_levels_workspace_bytes", _Z, [_I, _P] + [_I] * 5)
_proto("d2t_corr_bwd_levels_workspace_bytes", _Z, [_I, _P] + [_I] * 6)
SYMBOLS += ["d2t_corr_fwd_levels_f32", "d2t_corr_bwd_levels_f32", "d2t_corr_fwd_levels_workspace_bytes",
            "d2t_corr_bwd_levels_workspace_bytes"]
LAYOUT_REFERENCE, LAYOUT_CHANNEL_MAJOR = 0, 1
# (A, max_dets) ; (anchors, offsets, confs, A, conf_thresh, max_dets, iou_thresh, out_boxes, out_conf, out_idx, out_count, ws, ws_bytes, stream)
_proto("d2t_region_filter_workspace_bytes", _Z, [_I, _I])
_proto("d2t_region_filter_f32", _I, [_P, _P, _P, _I, ctypes.c_float, _I, ctypes.c_float, _P, _P, _P, _P, _P, _Z, _P])
# (anchors, offsets, confs, N, A, conf_thresh, max_dets, iou_thresh, out_boxes, out_conf, out_idx, out_count, ws, ws_bytes, stream)
_proto("d2t_region_filter_batched_f32", _I, [_P, _P, _P, _I, _I, ctypes.c_float, _I, ctypes.c_float, _P, _P, _P, _P, _P, _Z, _P])
SYMBOLS += ["d2t_region_filter_workspace_bytes", "d2t_region_filter_f32", "d2t_region_filter_batched_f32"]
_proto("d2t_psroipool_channels", _I, [_P, _I, _I, _P])
_proto("d2t_corr_mask", _I, [_P, _I, _I, _I, _I, _P])
SYMBOLS += ["d2t_psroipool_channels", "d2t_corr_mask"]
# the lab build (csrc/Makefile `lab`) exports one more symbol; the product library does not
IS_LAB_BUILD = hasattr(lib, "d2t_lab_build")


def check(code: int, what: str) -> None:
    """Turn a non-zero return code of the C ABI into the exception the reference would raise
    (its AT_ASSERTM / launch failures surface as RuntimeError)."""
    if code != 0:
        raise RuntimeError(f"{what} failed: {error_string(code).decode()} (code {code})")
