"""Region proposals on the device: d2t_region_filter_f32 against the numpy restatement of the host
pipeline the reference runs between the RPN and the R-FCN heads (oracle/regions.py; trainer.py:178-190)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "oracle"))
import regions as oracle_regions  # noqa: E402

DEV = "cuda:0"


def _anchors(h, w, rng):
    """reference utils.py:114-163 (build_anchors) with the default areas / aspect ratios (cfg/default.yaml:13-14)."""
    areas, ratios = [0.001, 0.004, 0.016, 0.064, 0.256], [0.5, 1.0, 2.0]
    dims = np.array([[np.sqrt(a * r), a / np.sqrt(a * r)] for a in areas for r in ratios])
    iv, jv = np.meshgrid(np.linspace(0, 1, h, endpoint=False) + 1 / h / 2, np.linspace(0, 1, w, endpoint=False) + 1 / w / 2, indexing="ij")
    ij = np.broadcast_to(np.stack([iv, jv], -1)[:, :, None, :], (h, w, len(dims), 2))
    hw = np.broadcast_to(dims[None, None], (h, w, len(dims), 2))
    return np.concatenate([ij, hw], 3).reshape(-1, 4).astype(np.float32)


def test_oracle_decode_and_nms_known_answers():
    a = np.array([[0.5, 0.5, 0.2, 0.4]], np.float32)
    np.testing.assert_allclose(oracle_regions.box_decode(a, np.array([[0.5, -0.25, 0.0, np.log(2.0)]], np.float32)),
                               [[0.6, 0.4, 0.2, 0.8]], rtol=1e-6)
    boxes = np.array([[0.5, 0.5, 0.2, 0.2], [0.5, 0.52, 0.2, 0.2], [0.1, 0.1, 0.1, 0.1], [0.5, 0.5, 0.2, 0.2]], np.float32)
    confs = np.array([0.9, 0.8, 0.7, 0.2], np.float32)
    keep, kb = oracle_regions.region_filter(confs, boxes, 0.3, 10, 0.5)
    assert keep.tolist() == [0, 2] and kb.shape == (2, 4)              # 1 overlaps 0 (IoU 0.82), 3 is under the threshold
    assert oracle_regions.region_filter(confs, boxes, 0.3, 1, 0.5)[0].tolist() == [0]


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(38, 75, 0.3, 3000, 0.5, 0.5), (38, 63, 0.3, 3000, 0.3, 0.5), (10, 12, 0.5, 64, 0.5, 0.5),
                                  (38, 75, 0.0, 4096, 0.7, 0.5), (20, 20, 0.9, 300, 0.5, 0.2), (5, 5, 0.99, 16, 0.5, 0.5)], ids=str)
def test_matches_numpy_pipeline(case):
    from detect_to_track.models import _ext
    h, w, thr, k, iou, spread = case
    rng = np.random.default_rng(h * 1000 + w)
    anchors = _anchors(h, w, rng)
    A = len(anchors)
    offsets = (rng.standard_normal((A, 4)) * spread * [0.5, 0.5, 0.3, 0.3]).astype(np.float32)
    confs = rng.random(A).astype(np.float32)
    confs[rng.integers(0, A, A // 50)] = confs[0]                      # ties across the top-k boundary
    boxes_d, conf_d, idx_d, count = _ext.region_filter(*(torch.from_numpy(x).to(DEV) for x in (anchors, offsets, confs)), thr, k, iou)
    n = int(count.item())
    # decode: compared on its own (expf vs numpy's exp: 1 ulp); the filters are compared on the device's own boxes
    want_boxes = oracle_regions.box_decode(anchors, offsets)
    idx = idx_d.cpu().numpy()
    assert (idx[:n] >= 0).all() and (idx[n:] == -1).all()
    got_boxes = boxes_d.cpu().numpy()
    np.testing.assert_allclose(got_boxes[:n], want_boxes[idx[:n]], rtol=2e-6, atol=1e-7)
    assert not got_boxes[n:].any() and not conf_d.cpu().numpy()[n:].any()
    all_boxes = want_boxes.copy()
    all_boxes[idx[:n]] = got_boxes[:n]                                 # the device's values where it reported them
    _check_every_nms_decision(confs, all_boxes, thr, k, iou, idx[:n], conf_d.cpu().numpy()[:n])


def _check_every_nms_decision(confs, boxes, thr, k, iou, dev_keep, dev_conf, tol=1e-6, max_marginal=4):
    """Replays the greedy NMS candidate by candidate FOLLOWING the device's decisions and checks each one: candidate c
    (in descending confidence, ties by anchor index) must be kept iff no earlier kept box has IoU > iou with it.  The only
    excuse is a decision whose own deciding IoU lies within `tol` of the threshold (the boxes of candidates the device
    dropped are not exported, and expf differs from numpy's exp in the last bit) -- and at most `max_marginal` of them.
    A wrong drop, a wrong keep, a wrong order or a wrong top-k cut anywhere else fails."""
    sel = np.nonzero(confs > np.float32(thr))[0]
    order = sel[np.argsort(-confs[sel], kind="stable")][:k]            # ConfidenceFilter + MaxDetFilter (ties: lower index first)
    dev_keep = np.asarray(dev_keep, dtype=np.int64)
    assert np.isin(dev_keep, order).all(), "the device kept a box that is not among the top-k candidates above the threshold"
    pos = {int(a): p for p, a in enumerate(order)}
    assert (np.diff([pos[int(a)] for a in dev_keep]) > 0).all(), "kept boxes are not in descending-confidence order"
    np.testing.assert_array_equal(dev_conf, confs[dev_keep])
    kept_set = set(int(a) for a in dev_keep)
    kept_boxes = np.empty((0, 4), np.float32)
    marginal = 0
    for a in order:
        b = boxes[a]
        ious = oracle_regions.iou_one_to_many(b, kept_boxes) if len(kept_boxes) else np.zeros(0, np.float32)
        top = float(ious.max()) if len(ious) else 0.0
        want_keep = not (ious > np.float32(iou)).any()
        got_keep = int(a) in kept_set
        if want_keep != got_keep:
            assert abs(top - iou) < tol, (f"candidate {int(a)}: device {'kept' if got_keep else 'dropped'} it, max IoU with the kept boxes "
                                          f"before it is {top:.7f} against a threshold of {iou}")
            marginal += 1
        if got_keep:
            kept_boxes = np.concatenate([kept_boxes, b[None]])
    assert marginal <= max_marginal, f"{marginal} decisions sat within {tol} of the threshold: not credible"


@pytest.mark.gpu
@pytest.mark.parametrize("k", [300, 3000], ids=["fused", "mask+nms"])
def test_batched_frames_equal_single_calls(k):
    """d2t_region_filter_batched_f32: the N frames of a step in one call; frame f's slices must be the single-frame call's
    results bit for bit (same kernels, blockIdx.z = frame)."""
    from detect_to_track.models import _ext
    rng = np.random.default_rng(11)
    anchors = _anchors(38, 63, rng)
    A, N = len(anchors), 4
    offsets = (rng.standard_normal((N, A, 4)) * 0.5 * [0.5, 0.5, 0.3, 0.3]).astype(np.float32)
    confs = rng.random((N, A)).astype(np.float32)
    confs[3] = 0.1                                                     # a frame with nothing above the threshold
    ta, to, tc = (torch.from_numpy(x).to(DEV) for x in (anchors, offsets, confs))
    bb, bc, bi, bn = _ext.region_filter_batched(ta, to, tc, 0.3, k, 0.5)
    assert bb.shape == (N, k, 4) and bn.shape == (N,)
    for f in range(N):
        sb, sc, si, sn = _ext.region_filter(ta, to[f].contiguous(), tc[f].contiguous(), 0.3, k, 0.5)
        assert torch.equal(bb[f], sb) and torch.equal(bc[f], sc) and torch.equal(bi[f], si) and int(bn[f]) == int(sn)
    assert int(bn[3]) == 0 and int(bn[0]) > 0


def test_nms_decision_checker_bites():
    """The checker itself (CPU): it accepts the numpy pipeline's own result and rejects a list with one box wrongly dropped,
    one wrongly kept, two swapped, or one from below the top-k cut."""
    rng = np.random.default_rng(3)
    anchors = _anchors(20, 20, rng)
    A = len(anchors)
    offsets = (rng.standard_normal((A, 4)) * 0.5 * [0.5, 0.5, 0.3, 0.3]).astype(np.float32)
    confs = rng.random(A).astype(np.float32)
    boxes = oracle_regions.box_decode(anchors, offsets)
    keep, _ = oracle_regions.region_filter(confs, boxes, 0.3, 300, 0.5)
    _check_every_nms_decision(confs, boxes, 0.3, 300, 0.5, keep, confs[keep])
    sel = np.nonzero(confs > np.float32(0.3))[0]
    order = sel[np.argsort(-confs[sel], kind="stable")]
    dropped = [a for a in order[:300] if a not in set(keep.tolist())]
    wrong = {"drop": np.delete(keep, 7), "keep": np.sort(np.append(keep, dropped[0]))[::1], "swap": keep[[1, 0] + list(range(2, len(keep)))],
             "below_cut": np.append(keep, order[300])}
    wrong["keep"] = np.asarray(sorted(wrong["keep"].tolist(), key=lambda a: list(order).index(a)))
    for name, bad in wrong.items():
        with pytest.raises(AssertionError):
            _check_every_nms_decision(confs, boxes, 0.3, 300, 0.5, bad, confs[bad])


@pytest.mark.gpu
def test_degenerate_inputs():
    from detect_to_track.models import _ext
    a = torch.tensor(_anchors(4, 4, None), device=DEV)
    A = a.shape[0]
    off = torch.zeros(A, 4, device=DEV)
    # nothing above the threshold: count 0, all padding
    b, c, i, n = _ext.region_filter(a, off, torch.full((A,), 0.1, device=DEV), 0.3, 32, 0.5)
    assert int(n) == 0 and not b.any() and bool((i == -1).all())
    # NaN confidences are filtered out; identical boxes collapse to the first one of each anchor shape
    conf = torch.full((A,), 0.9, device=DEV)
    conf[::3] = float("nan")
    b, c, i, n = _ext.region_filter(a, off, conf, 0.3, 32, 0.5)
    kept = i[: int(n)].cpu().numpy()
    assert int(n) > 0 and (kept % 3 != 0).all() and (np.diff(kept) > 0).all()       # equal confidences: anchor order
    with pytest.raises(RuntimeError):
        _ext.region_filter(a, off, conf, 0.3, 5000, 0.5)                              # max_dets > 4096
    # boxes move as 16-byte vectors: a pointer that is only 4-byte aligned is refused (D2T_EINVAL), not dereferenced
    skew = torch.zeros(4 * A + 1, device=DEV)[1:].view(A, 4)
    assert skew.data_ptr() % 16 == 4 and skew.is_contiguous()
    with pytest.raises(RuntimeError):
        _ext.region_filter(skew, off, conf, 0.3, 32, 0.5)
