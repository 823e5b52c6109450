"""numpy restatement of the host-side region pipeline between the RPN and the R-FCN heads (TEST INFRASTRUCTURE:
imported by tests/ only).

  box_decode      /root/reference/detect_to_track/data/encoding.py:182-206 (frcnn_box_decode), float32
  region_filter   the composition /root/reference/detect_to_track/trainer.py:98-102,189-190 builds from three filters
                  of `ml_utils` (requirements.txt: ml-utils==3.0.0, NOT vendored with the reference and not installed
                  here).  PARITY UNPINNED: the filters are restated from their names and call sites as the standard
                  operations -- conf > thresh; the max_dets highest confidences (stable: lower index first on ties);
                  greedy NMS in descending confidence with IoU > thresh removing the later box.
Boxes are (centre_i, centre_j, height, width) fractions of the frame.
"""
import numpy as np


def box_decode(anchors: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    anchors, offsets = np.asarray(anchors, np.float32), np.asarray(offsets, np.float32)
    t_ij, t_hw = np.hsplit(offsets, 2)
    a_ij, a_hw = np.hsplit(anchors, 2)
    return np.concatenate([t_ij * a_hw + a_ij, np.exp(t_hw) * a_hw], axis=1).astype(np.float32)


def iou_one_to_many(box: np.ndarray, others: np.ndarray) -> np.ndarray:
    """float32, same expression order as the device kernel (so that threshold decisions agree bit for bit)."""
    f = np.float32
    two = f(2)
    ai0, ai1, aj0, aj1 = box[0] - box[2] / two, box[0] + box[2] / two, box[1] - box[3] / two, box[1] + box[3] / two
    bi0, bi1 = others[:, 0] - others[:, 2] / two, others[:, 0] + others[:, 2] / two
    bj0, bj1 = others[:, 1] - others[:, 3] / two, others[:, 1] + others[:, 3] / two
    ih = np.minimum(ai1, bi1) - np.maximum(ai0, bi0)
    iw = np.minimum(aj1, bj1) - np.maximum(aj0, bj0)
    inter = np.where(ih > 0, ih, f(0)) * np.where(iw > 0, iw, f(0))
    uni = box[2] * box[3] + others[:, 2] * others[:, 3] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(uni > 0, inter / uni, f(0)).astype(np.float32)


def region_filter(confs: np.ndarray, boxes: np.ndarray, conf_thresh: float, max_dets: int, iou_thresh: float):
    """Returns (kept anchor indices in descending confidence, their boxes)."""
    confs, boxes = np.asarray(confs, np.float32), np.asarray(boxes, np.float32)
    idx = np.nonzero(confs > np.float32(conf_thresh))[0]
    order = idx[np.argsort(-confs[idx], kind="stable")][:max_dets]
    b = boxes[order]
    removed = np.zeros(len(order), dtype=bool)
    keep = []
    for i in range(len(order)):
        if removed[i]:
            continue
        keep.append(i)
        if i + 1 < len(order):
            removed[i + 1:] |= iou_one_to_many(b[i], b[i + 1:]) > np.float32(iou_thresh)
    keep = np.asarray(keep, dtype=np.int64)
    return order[keep], b[keep]
