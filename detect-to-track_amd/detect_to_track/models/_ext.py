"""Tensor-level entry points with the names and argument order of the reference's pybind
module ``_ext`` (pointwise_correlation.cpp:51-62, roipool.cpp:48-59, ps_roipool.cpp:50-61):

    pointwise_correlation_forward(FM0, FM1, d_max, stride) -> out
    pointwise_correlation_backward(grad_out, FM0, FM1, d_max, stride) -> (grad_FM0, grad_FM1)
    roipool_forward(FM, rois, r_hw) -> out
    roipool_backward(grad_out, rois, i_h, i_w) -> grad_FM
    ps_roipool_forward(FM, rois, n_targets, r_hw) -> out
    ps_roipool_backward(grad_out, rois, i_h, i_w) -> grad_FM

Each validates its tensors the way the reference's CHECK_INPUT does (common/cpp_common.hpp:1-3:
device first, then contiguity; both RuntimeError), allocates the outputs with ``torch.empty``
(the kernels write every element, including the structural zeros the reference gets from
``at::zeros``), and launches the HIP kernels of libd2t_ops.so on the caller's current stream.
"""
import ctypes
from typing import Tuple

import torch
from torch import Tensor

from . import _native

_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}

# Leaving the tuned envelope is not silent: the first call of each kind says so once (measured on an MI355X at B = 8, C = 256,
# 38 x 63 / R = 300, C = 1024: tools/envelope_cost.py, profiles/r04_d_envelope_cost.jsonl).  float64 runs the type-generic
# kernels by design (the reference's gradcheck tests) and is not reported.
_ENVELOPE_WARNED = set()


def _outside_envelope(kind: str, why: str, factor: str) -> None:
    if kind in _ENVELOPE_WARNED:
        return
    _ENVELOPE_WARNED.add(kind)
    import warnings
    warnings.warn(f"detect_to_track {kind}: {why} is outside the gfx950-tuned kernels' envelope; reference-order kernels run "
                  f"instead (same results; at the model's shapes: {factor})", RuntimeWarning, stacklevel=3)


def _check_corr_envelope(x: Tensor, d_max: int, stride: int, impl: int) -> None:
    if x.dtype == torch.float32 and impl == _native.IMPL_AUTO and x.numel() and (d_max != 8 or stride != 1 or x.shape[-1] < 20):
        _outside_envelope("PointwiseCorrelation", f"d_max = {d_max}, stride = {stride}, W = {x.shape[-1]} (tuned: d_max = 8, stride 1, W >= 20)",
                          "about 2x (forward) / 2.7x (backward) slower")


def _check_corr_bwd_envelope(x: Tensor, d_max: int, stride: int, impl: int) -> None:
    """The backward's envelope is narrower than the forward's: the 8-wave strip kernel walks super-steps of 4 map rows with five tiles
    alive and needs H >= 17 (csrc/d2t_corr_bwd8.hip: corr_bwd8_supported, tiles_i >= 5; include/d2t_ops.h, D2T_IMPL_MFMA)."""
    if x.dtype == torch.float32 and impl == _native.IMPL_AUTO and x.numel() and d_max == 8 and stride == 1 and x.shape[-1] >= 20 and x.shape[-2] < 17:
        _outside_envelope("PointwiseCorrelation backward", f"H = {x.shape[-2]} (tuned backward: H >= 17)", "about 2.7x slower")


def _check_pool_envelope(kind: str, x: Tensor, k: int, impl: int) -> None:
    if x.dtype == torch.float32 and impl == _native.IMPL_AUTO and x.numel() and k != 7:
        if kind == "ROIPool":
            # the forward for r_hw <= 16 (and >= 32 RoIs) stays on the summed-area kernel: the SAME contract as r_hw = 7 (within 1e-5 of the
            # reference, NaN pattern exact -- not the reference-order kernel's bits); r_hw > 16 and every backward take reference-order kernels
            how = ("the forward keeps the summed-area kernel for r_hw <= 16 (within 1e-5 of the reference, as for r_hw = 7; 5x slower "
                   "reference-order kernel above), the backward runs reference-order kernels (same results) 2.5-4x slower")
        else:
            how = "reference-order kernels run instead (same results): the forward as fast, the backward 2.5-4x slower"
        if kind in _ENVELOPE_WARNED:
            return
        _ENVELOPE_WARNED.add(kind)
        import warnings
        warnings.warn(f"detect_to_track {kind}: r_hw = {k} (tuned: 7) is outside the gfx950-tuned kernels' envelope; {how}",
                      RuntimeWarning, stacklevel=3)


def _check_input(x: Tensor, name: str) -> None:
    if not isinstance(x, Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(x).__name__}")
    if not x.is_cuda:
        raise RuntimeError("CPU op not implemented")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _suffix(x: Tensor, name: str) -> str:
    try:
        return _SUFFIX[x.dtype]
    except KeyError:
        raise RuntimeError(f'"{name}" not implemented for \'{str(x.dtype).replace("torch.", "")}\'') from None


def _same(x: Tensor, ref: Tensor, xname: str, refname: str) -> None:
    if x.dtype != ref.dtype:
        raise RuntimeError(f"expected scalar type {ref.dtype} for {xname} (dtype of {refname}) but found {x.dtype}")
    if x.device != ref.device:
        raise RuntimeError(f"{xname} is on {x.device} but {refname} is on {ref.device}")


def _workspace(nbytes: int, like: Tensor):
    if nbytes == 0:
        return None, 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
    return ws, nbytes


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


# --------------------------------------------------------------------------- correlation
def pointwise_correlation_forward(FM0: Tensor, FM1: Tensor, d_max: int, stride: int,
                                  impl: int = _native.IMPL_AUTO) -> Tensor:
    _check_input(FM0, "FM0")
    _check_input(FM1, "FM1")
    sfx = _suffix(FM0, "pointwiseCorrelationsKernelForward")
    _same(FM1, FM0, "FM1", "FM0")
    if FM0.dim() != 4 or FM1.shape != FM0.shape:
        raise RuntimeError(f"FM0 and FM1 must both be (B, C, H, W); got {tuple(FM0.shape)} and {tuple(FM1.shape)}")
    d_max, stride = int(d_max), int(stride)
    _check_corr_envelope(FM0, d_max, stride, impl)
    B, C, H, W = FM0.shape
    cw = 2 * d_max + 1
    with torch.cuda.device(FM0.device):
        out = torch.empty((B, H, W, cw, cw), dtype=FM0.dtype, device=FM0.device)
        ws, n = _workspace(_native.lib.d2t_corr_fwd_workspace_bytes(B, C, H, W, d_max, stride, FM0.element_size()), FM0)
        rc = getattr(_native.lib, f"d2t_corr_fwd_{sfx}")(
            FM0.data_ptr(), FM1.data_ptr(), out.data_ptr(), B, C, H, W, d_max, stride,
            _ptr(ws), n, impl, _stream(FM0))
    _native.check(rc, "pointwise_correlation_forward")
    return out


def pointwise_correlation_backward(grad_out: Tensor, FM0: Tensor, FM1: Tensor, d_max: int, stride: int,
                                   impl: int = _native.IMPL_AUTO) -> Tuple[Tensor, Tensor]:
    _check_input(grad_out, "gradOut")
    _check_input(FM0, "FM0")
    _check_input(FM1, "FM1")
    sfx = _suffix(FM0, "pointwiseCorrelationsKernelBackward")
    _same(FM1, FM0, "FM1", "FM0")
    _same(grad_out, FM0, "gradOut", "FM0")
    d_max, stride = int(d_max), int(stride)
    B, C, H, W = FM0.shape
    cw = 2 * d_max + 1
    if FM1.shape != FM0.shape or tuple(grad_out.shape) != (B, H, W, cw, cw):
        raise RuntimeError(
            f"shape mismatch: FM0 {tuple(FM0.shape)}, FM1 {tuple(FM1.shape)}, gradOut {tuple(grad_out.shape)} "
            f"(expected gradOut {(B, H, W, cw, cw)})")
    _check_corr_bwd_envelope(FM0, d_max, stride, impl)
    with torch.cuda.device(FM0.device):
        g0 = torch.empty_like(FM0)
        g1 = torch.empty_like(FM1)
        ws, n = _workspace(_native.lib.d2t_corr_bwd_workspace_bytes(B, C, H, W, d_max, stride, FM0.element_size()), FM0)
        rc = getattr(_native.lib, f"d2t_corr_bwd_{sfx}")(
            grad_out.data_ptr(), FM0.data_ptr(), FM1.data_ptr(), g0.data_ptr(), g1.data_ptr(),
            B, C, H, W, d_max, stride, _ptr(ws), n, impl, _stream(FM0))
    _native.check(rc, "pointwise_correlation_backward")
    return g0, g1


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def pointwise_correlation_levels_forward(FM0s, FM1s, d_max: int, stride: int, out=None,
                                         impl: int = _native.IMPL_AUTO):
    """Correlate several feature-map pairs that share (B, H, W) in one call (one launch for small
    grids) and return / fill CHANNEL-MAJOR outputs: level l gives ((2d+1)^2, H, W) per batch item --
    the tensor reference correlation_tracker.py:64-70 builds with view + permute.

    out: None -> a fresh (B, L*(2d+1)^2, H, W) float32 tensor is returned, level l in channels
    [l*(2d+1)^2, (l+1)*(2d+1)^2); or a contiguous (B, Cout, H, W) tensor plus a first channel
    ``(tensor, c0)``: level l is written at channels c0 + l*(2d+1)^2 (a torch.cat target)."""
    L = len(FM0s)
    if L < 1 or L != len(FM1s):
        raise RuntimeError("need as many FM1 as FM0 feature maps (at least one)")
    for k, (a, b) in enumerate(zip(FM0s, FM1s)):
        _check_input(a, f"FM0[{k}]")
        _check_input(b, f"FM1[{k}]")
        _same(b, a, f"FM1[{k}]", f"FM0[{k}]")
        _same(a, FM0s[0], f"FM0[{k}]", "FM0[0]")
        if a.dtype != torch.float32:
            raise RuntimeError("the fused correlation path is float32 only")
        if a.dim() != 4 or b.shape != a.shape or a.shape[0] != FM0s[0].shape[0] or a.shape[2:] != FM0s[0].shape[2:]:
            raise RuntimeError("every level must be (B, C_l, H, W) with the same B, H, W")
    d_max, stride = int(d_max), int(stride)
    B, _, H, W = FM0s[0].shape
    cells = (2 * d_max + 1) ** 2
    dev = FM0s[0].device
    with torch.cuda.device(dev):
        if out is None:
            buf, c0 = torch.empty((B, L * cells, H, W), dtype=torch.float32, device=dev), 0
        else:
            buf, c0 = out
            _check_input(buf, "out")
            if buf.dtype != torch.float32 or buf.dim() != 4 or buf.shape[0] != B or tuple(buf.shape[2:]) != (H, W) \
                    or c0 < 0 or c0 + L * cells > buf.shape[1]:
                raise RuntimeError(f"out must be float32 (B, >= {c0 + L * cells}, H, W), got {tuple(buf.shape)}")
        outs = [buf[:, c0 + l * cells: c0 + (l + 1) * cells] for l in range(L)]
        Cs = (ctypes.c_int * L)(*[int(a.shape[1]) for a in FM0s])
        ws, n = _workspace(_native.lib.d2t_corr_fwd_levels_workspace_bytes(L, Cs, B, H, W, d_max, stride), FM0s[0])
        rc = _native.lib.d2t_corr_fwd_levels_f32(
            L, _ptr_array(FM0s), _ptr_array(FM1s), _ptr_array(outs), Cs, B, H, W, d_max, stride,
            _native.LAYOUT_CHANNEL_MAJOR, buf.shape[1] * H * W, _ptr(ws), n, impl, _stream(FM0s[0]))
    _native.check(rc, "pointwise_correlation_levels_forward")
    return buf


def pointwise_correlation_levels_backward(grad, c0: int, FM0s, FM1s, d_max: int, stride: int,
                                          impl: int = _native.IMPL_AUTO):
    """Gradients of pointwise_correlation_levels_forward: grad is the contiguous (B, Cout, H, W) gradient
    of the channel-major buffer, level l's cells start at channel c0 + l*(2d+1)^2.  Returns two lists."""
    _check_input(grad, "gradOut")
    L = len(FM0s)
    d_max, stride = int(d_max), int(stride)
    B, _, H, W = FM0s[0].shape
    cells = (2 * d_max + 1) ** 2
    if grad.dtype != torch.float32 or grad.dim() != 4 or grad.shape[0] != B or tuple(grad.shape[2:]) != (H, W) \
            or c0 + L * cells > grad.shape[1]:
        raise RuntimeError(f"gradOut must be float32 (B, >= {c0 + L * cells}, H, W), got {tuple(grad.shape)}")
    with torch.cuda.device(grad.device):
        g0 = [torch.empty_like(a) for a in FM0s]
        g1 = [torch.empty_like(b) for b in FM1s]
        gouts = [grad[:, c0 + l * cells: c0 + (l + 1) * cells] for l in range(L)]
        Cs = (ctypes.c_int * L)(*[int(a.shape[1]) for a in FM0s])
        ws, n = _workspace(_native.lib.d2t_corr_bwd_levels_workspace_bytes(L, Cs, B, H, W, d_max, stride, _native.LAYOUT_CHANNEL_MAJOR), grad)
        rc = _native.lib.d2t_corr_bwd_levels_f32(
            L, _ptr_array(gouts), _ptr_array(FM0s), _ptr_array(FM1s), _ptr_array(g0), _ptr_array(g1), Cs,
            B, H, W, d_max, stride, _native.LAYOUT_CHANNEL_MAJOR, grad.shape[1] * H * W, _ptr(ws), n, impl, _stream(grad))
    _native.check(rc, "pointwise_correlation_levels_backward")
    return g0, g1


# --------------------------------------------------------------------------- roipool
def _check_rois(rois: Tensor, like: Tensor, likename: str) -> int:
    _check_input(rois, "rois")
    _same(rois, like, "rois", likename)
    if rois.dim() != 2 or rois.size(1) != 4:
        raise RuntimeError(f"rois must be (|R|, 4) ijhw fractional boxes, got {tuple(rois.shape)}")
    return rois.size(0)


def roipool_forward(FM: Tensor, rois: Tensor, r_hw: int, impl: int = _native.IMPL_AUTO) -> Tensor:
    _check_input(FM, "FM")
    sfx = _suffix(FM, "ROIPoolKernelForward")
    R = _check_rois(rois, FM, "FM")
    if FM.dim() != 3:
        raise RuntimeError(f"FM must be (C, H, W), got {tuple(FM.shape)}")
    r_hw = int(r_hw)
    _check_pool_envelope("ROIPool", FM, r_hw, impl)
    C, H, W = FM.shape
    with torch.cuda.device(FM.device):
        out = torch.empty((R, C, r_hw, r_hw), dtype=FM.dtype, device=FM.device)
        ws, n = _workspace(_native.lib.d2t_roipool_fwd_workspace_bytes(R, C, H, W, r_hw, FM.element_size()), FM)
        rc = getattr(_native.lib, f"d2t_roipool_fwd_{sfx}")(
            FM.data_ptr(), rois.data_ptr(), out.data_ptr(), R, C, H, W, r_hw, _ptr(ws), n, impl, _stream(FM))
    _native.check(rc, "roipool_forward")
    return out


def roipool_backward(grad_out: Tensor, rois: Tensor, i_h: int, i_w: int, impl: int = _native.IMPL_AUTO) -> Tensor:
    _check_input(grad_out, "gradOut")
    sfx = _suffix(grad_out, "ROIPoolKernelBackward")
    R = _check_rois(rois, grad_out, "gradOut")
    if grad_out.dim() != 4 or grad_out.size(0) != R or grad_out.size(2) != grad_out.size(3):
        raise RuntimeError(f"gradOut must be (|R|, C, k, k) with |R| = {R}, got {tuple(grad_out.shape)}")
    _, C, k, _ = grad_out.shape
    H, W = int(i_h), int(i_w)
    with torch.cuda.device(grad_out.device):
        gin = torch.empty((C, H, W), dtype=grad_out.dtype, device=grad_out.device)
        ws, n = _workspace(_native.lib.d2t_roipool_bwd_workspace_bytes(R, C, H, W, k, grad_out.element_size()), grad_out)
        rc = getattr(_native.lib, f"d2t_roipool_bwd_{sfx}")(
            grad_out.data_ptr(), rois.data_ptr(), gin.data_ptr(), R, C, H, W, k, _ptr(ws), n, impl, _stream(grad_out))
    _native.check(rc, "roipool_backward")
    return gin


# --------------------------------------------------------------------------- psroipool
def ps_roipool_forward(FM: Tensor, rois: Tensor, n_targets: int, r_hw: int, impl: int = _native.IMPL_AUTO) -> Tensor:
    _check_input(FM, "FM")
    sfx = _suffix(FM, "psROIPoolKernelForward")
    R = _check_rois(rois, FM, "FM")
    n_targets, r_hw = int(n_targets), int(r_hw)
    if FM.dim() != 3 or FM.size(0) != n_targets * r_hw * r_hw:
        # the reference's launcher never checks this (ps_roipool_cuda.cu:144-174) and would read
        # out of bounds; the Function raises ValueError before reaching here (ps_roipool.py:44-49)
        raise RuntimeError(f"FM must be ({n_targets * r_hw * r_hw}, H, W), got {tuple(FM.shape)}")
    _check_pool_envelope("PSROIPool", FM, r_hw, impl)
    _, H, W = FM.shape
    with torch.cuda.device(FM.device):
        out = torch.empty((R, n_targets, r_hw, r_hw), dtype=FM.dtype, device=FM.device)
        ws, n = _workspace(_native.lib.d2t_psroipool_fwd_workspace_bytes(R, n_targets, H, W, r_hw, FM.element_size()), FM)
        rc = getattr(_native.lib, f"d2t_psroipool_fwd_{sfx}")(
            FM.data_ptr(), rois.data_ptr(), out.data_ptr(), R, n_targets, H, W, r_hw, _ptr(ws), n, impl, _stream(FM))
    _native.check(rc, "ps_roipool_forward")
    return out


def ps_roipool_backward(grad_out: Tensor, rois: Tensor, i_h: int, i_w: int, impl: int = _native.IMPL_AUTO) -> Tensor:
    _check_input(grad_out, "gradOut")
    sfx = _suffix(grad_out, "psROIPoolKernelBackward")
    R = _check_rois(rois, grad_out, "gradOut")
    if grad_out.dim() != 4 or grad_out.size(0) != R or grad_out.size(2) != grad_out.size(3):
        raise RuntimeError(f"gradOut must be (|R|, nT, k, k) with |R| = {R}, got {tuple(grad_out.shape)}")
    _, nT, k, _ = grad_out.shape
    H, W = int(i_h), int(i_w)
    with torch.cuda.device(grad_out.device):
        gin = torch.empty((nT * k * k, H, W), dtype=grad_out.dtype, device=grad_out.device)
        ws, n = _workspace(_native.lib.d2t_psroipool_bwd_workspace_bytes(R, nT, H, W, k, grad_out.element_size()), grad_out)
        rc = getattr(_native.lib, f"d2t_psroipool_bwd_{sfx}")(
            grad_out.data_ptr(), rois.data_ptr(), gin.data_ptr(), R, nT, H, W, k, _ptr(ws), n, impl, _stream(grad_out))
    _native.check(rc, "ps_roipool_backward")
    return gin


# --------------------------------------------------------------------------- introspection
def roipool_bins(rois: Tensor, i_h: int, i_w: int, r_hw: int, position_sensitive: bool = False) -> Tensor:
    """(|R|, k, k, 4) int32 pixel bounds {i0, i1, j0, j1} computed by the device code the
    pooling kernels use (bit-exact parity of index arithmetic)."""
    _check_input(rois, "rois")
    sfx = _suffix(rois, "bins")
    R = rois.size(0)
    op = "psroipool" if position_sensitive else "roipool"
    with torch.cuda.device(rois.device):
        out = torch.empty((R, r_hw, r_hw, 4), dtype=torch.int32, device=rois.device)
        rc = getattr(_native.lib, f"d2t_{op}_bins_{sfx}")(
            rois.data_ptr(), out.data_ptr(), R, int(i_h), int(i_w), int(r_hw), _stream(rois))
    _native.check(rc, f"{op}_bins")
    return out


def ps_roipool_channels(n_targets: int, r_hw: int, device) -> Tensor:
    dev = torch.device(device)
    with torch.cuda.device(dev):
        out = torch.empty((n_targets, r_hw, r_hw), dtype=torch.int32, device=dev)
        rc = _native.lib.d2t_psroipool_channels(out.data_ptr(), int(n_targets), int(r_hw),
                                                torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "psroipool_channels")
    return out


def pointwise_correlation_mask(i_h: int, i_w: int, d_max: int, stride: int, device) -> Tensor:
    dev = torch.device(device)
    cw = 2 * int(d_max) + 1
    with torch.cuda.device(dev):
        out = torch.empty((i_h, i_w, cw, cw), dtype=torch.uint8, device=dev)
        rc = _native.lib.d2t_corr_mask(out.data_ptr(), int(i_h), int(i_w), int(d_max), int(stride),
                                       torch.cuda.current_stream(dev).cuda_stream)
    _native.check(rc, "corr_mask")
    return out


def region_filter(anchors: Tensor, offsets: Tensor, confs: Tensor, conf_thresh: float, max_dets: int, iou_thresh: float):
    """RPN outputs -> region proposals without leaving the device (reference trainer.py:178-190: numpy
    frcnn_box_decode + ConfidenceFilter + MaxDetFilter + NMSFilter on host copies).

    anchors, offsets (A,4), confs (A,): float32 CUDA.  Returns (boxes (max_dets,4), confs (max_dets,),
    anchor_index (max_dets,) int32, count (1,) int32): survivors first, in descending confidence, then padding
    (zero boxes, index -1).  Nothing is read back: shapes are static."""
    for t, name in ((anchors, "anchors"), (offsets, "offsets"), (confs, "confs")):
        _check_input(t, name)
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} must be float32")
        _same(t, anchors, name, "anchors")
    A = int(confs.numel())
    if tuple(anchors.shape) != (A, 4) or tuple(offsets.shape) != (A, 4):
        raise RuntimeError(f"anchors / offsets must be ({A}, 4), got {tuple(anchors.shape)} / {tuple(offsets.shape)}")
    max_dets = int(max_dets)
    dev = anchors.device
    with torch.cuda.device(dev):
        boxes = torch.empty((max_dets, 4), dtype=torch.float32, device=dev)
        conf = torch.empty((max_dets,), dtype=torch.float32, device=dev)
        idx = torch.empty((max_dets,), dtype=torch.int32, device=dev)
        count = torch.empty((1,), dtype=torch.int32, device=dev)
        ws, n = _workspace(_native.lib.d2t_region_filter_workspace_bytes(A, max_dets), anchors)
        rc = _native.lib.d2t_region_filter_f32(anchors.data_ptr(), offsets.data_ptr(), confs.data_ptr(), A, float(conf_thresh),
                                               max_dets, float(iou_thresh), boxes.data_ptr(), conf.data_ptr(), idx.data_ptr(),
                                               count.data_ptr(), _ptr(ws), n, _stream(anchors))
    _native.check(rc, "region_filter")
    return boxes, conf, idx, count


def region_filter_batched(anchors: Tensor, offsets: Tensor, confs: Tensor, conf_thresh: float, max_dets: int, iou_thresh: float):
    """``region_filter`` for the N frames of a step in ONE call of the library (the reference runs its host pipeline once per
    frame, trainer.py:178-207): anchors (A,4) shared, offsets (N,A,4), confs (N,A) -> (boxes (N,max_dets,4), confs (N,max_dets),
    anchor_index (N,max_dets) int32, count (N,) int32); frame f's slices equal a single-frame call's results bit for bit."""
    for t, name in ((anchors, "anchors"), (offsets, "offsets"), (confs, "confs")):
        _check_input(t, name)
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} must be float32")
        _same(t, anchors, name, "anchors")
    if confs.dim() != 2:
        raise RuntimeError(f"confs must be (N, A), got {tuple(confs.shape)}")
    N, A = int(confs.shape[0]), int(confs.shape[1])
    if tuple(anchors.shape) != (A, 4) or tuple(offsets.shape) != (N, A, 4):
        raise RuntimeError(f"anchors / offsets must be ({A}, 4) / ({N}, {A}, 4), got {tuple(anchors.shape)} / {tuple(offsets.shape)}")
    max_dets = int(max_dets)
    dev = anchors.device
    with torch.cuda.device(dev):
        boxes = torch.empty((N, max_dets, 4), dtype=torch.float32, device=dev)
        conf = torch.empty((N, max_dets), dtype=torch.float32, device=dev)
        idx = torch.empty((N, max_dets), dtype=torch.int32, device=dev)
        count = torch.empty((N,), dtype=torch.int32, device=dev)
        ws, n = _workspace(N * _native.lib.d2t_region_filter_workspace_bytes(A, max_dets), anchors)
        rc = _native.lib.d2t_region_filter_batched_f32(anchors.data_ptr(), offsets.data_ptr(), confs.data_ptr(), N, A, float(conf_thresh),
                                                       max_dets, float(iou_thresh), boxes.data_ptr(), conf.data_ptr(), idx.data_ptr(),
                                                       count.data_ptr(), _ptr(ws), n, _stream(anchors))
    _native.check(rc, "region_filter_batched")
    return boxes, conf, idx, count
