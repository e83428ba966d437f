"""Instance-segmentation agreement metrics used as parity gates when two pipelines see different inputs
(SURVEY §8d "Parity gates") and by the evaluation callers (§8 f4).

Restates the behaviour of the reference's panoptic quality (`cell_segmentation/utils/metrics.py:41-150`,
`get_fast_pq` / `remap_label`) with a different mechanism: instead of materialising one mask per instance, the
pairwise intersections come from ONE joint histogram of (true id, pred id) pairs, so the cost is O(pixels) rather
than O(instances x pixels).  Pinned to outputs of the imported reference in `tests/golden/pq_cases.npz`
(`tools/make_golden_pq.py`).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def remap_label(inst: np.ndarray, by_size: bool = False) -> np.ndarray:
    """Contiguous ids 1..n (0 = background).  Order of first appearance in sorted-id order is kept unless
    `by_size`, in which case larger instances get smaller ids (ties: smaller original id first)
    (reference `metrics.py:155-184`)."""
    ids, inv, counts = np.unique(inst, return_inverse=True, return_counts=True)
    fg = ids != 0
    order = np.flatnonzero(fg)
    if by_size:
        order = order[np.argsort(-counts[order], kind="stable")]
    lut = np.zeros(len(ids), dtype=np.int32)
    lut[order] = np.arange(1, len(order) + 1, dtype=np.int32)
    return lut[inv].reshape(inst.shape)


def pairwise_iou(true: np.ndarray, pred: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """IoU matrix [n_true, n_pred] between the instances of two label maps, plus the sorted foreground id lists."""
    t_ids, t_inv = np.unique(true, return_inverse=True)
    p_ids, p_inv = np.unique(pred, return_inverse=True)
    nt, npd = len(t_ids), len(p_ids)
    joint = np.bincount(t_inv.ravel().astype(np.int64) * npd + p_inv.ravel(), minlength=nt * npd).reshape(nt, npd)
    t_area = joint.sum(1, keepdims=True).astype(np.float64)
    p_area = joint.sum(0, keepdims=True).astype(np.float64)
    inter = joint.astype(np.float64)
    union = t_area + p_area - inter
    with np.errstate(invalid="ignore", divide="ignore"):
        iou = np.where(inter > 0, inter / union, 0.0)
    t_fg, p_fg = t_ids != 0, p_ids != 0
    return iou[np.ix_(t_fg, p_fg)], t_ids[t_fg], p_ids[p_fg]


def panoptic_quality(true: np.ndarray, pred: np.ndarray, match_iou: float = 0.5):
    """[dq, sq, pq], [paired_true, paired_pred, unpaired_true, unpaired_pred] with the reference's conventions:
    ids are the (contiguous) label values, a pair needs IoU > `match_iou`; for thresholds below 0.5 the pairing is the
    maximum-weight assignment; dq = tp / (tp + fp/2 + fn/2 + 1e-6), sq = sum IoU / (tp + 1e-6)."""
    if match_iou < 0.0:
        raise AssertionError("Cant' be negative")
    iou, t_ids, p_ids = pairwise_iou(np.asarray(true), np.asarray(pred))
    if match_iou >= 0.5:
        ti, pi = np.nonzero(iou > match_iou)
        paired_iou = iou[ti, pi]
    else:
        from scipy.optimize import linear_sum_assignment
        ti, pi = linear_sum_assignment(-iou)
        sel = iou[ti, pi] > match_iou
        ti, pi = ti[sel], pi[sel]
        paired_iou = iou[ti, pi]
    paired_true = [int(v) for v in t_ids[ti]]
    paired_pred = [int(v) for v in p_ids[pi]]
    pt, pp = set(paired_true), set(paired_pred)
    unpaired_true = [int(v) for v in t_ids if int(v) not in pt]
    unpaired_pred = [int(v) for v in p_ids if int(v) not in pp]
    tp, fp, fn = len(paired_true), len(unpaired_pred), len(unpaired_true)
    dq = tp / (tp + 0.5 * fp + 0.5 * fn + 1.0e-6)
    sq = float(paired_iou.sum()) / (tp + 1.0e-6)
    return [dq, sq, dq * sq], [paired_true, paired_pred, unpaired_true, unpaired_pred]


def binary_pq_batch(true_maps: List[np.ndarray], pred_maps: List[np.ndarray]) -> float:
    """Mean bPQ over tiles (ids are remapped first, as the evaluation callers do,
    `inference_cellvit_experiment_pannuke.py:650-660`)."""
    vals = [panoptic_quality(remap_label(t), remap_label(p))[0][2] for t, p in zip(true_maps, pred_maps)]
    return float(np.mean(vals)) if vals else 0.0
