"""Instance-segmentation agreement metrics used as parity gates when two pipelines see different inputs
(SURVEY §8d "Parity gates") and by the evaluation callers (§8 f4).

Restates the behaviour of the reference's panoptic quality (`cell_segmentation/utils/metrics.py:41-150`,
`get_fast_pq` / `remap_label`) with a different mechanism: instead of materialising one mask per instance, the
pairwise intersections come from ONE joint histogram of (true id, pred id) pairs, so the cost is O(pixels) rather
than O(instances x pixels).  Pinned to outputs of the imported reference in `tests/golden/pq_cases.npz`
(`tools/make_golden_pq.py`) and, for `binarize` and the detection scores, `tests/golden/eval_cases.npz`
(`tools/make_golden_eval.py`).  `binary_dice` / `binary_jaccard` restate the torchmetrics calls of the evaluation caller
(torchmetrics is not installed here: parity unpinned, the formulas are the library's documented ones);
`pair_coordinates` is the reference's scipy call sequence (its module needs numba to import, so it is pinned by an
exhaustive-assignment property test instead).
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


def binarize(x: np.ndarray) -> np.ndarray:
    """Per-class instance maps [H, W, C] -> one instance map with ids 1..n in (channel, ascending id) order; where
    instances of different channels overlap the later one wins (reference `metrics.py:190-211`, which multiplies the
    running map by the complement of every instance mask; here one lookup per channel)."""
    x = np.asarray(x)
    out = np.zeros(x.shape[:2], dtype=np.int64)
    count = 0
    for c in range(x.shape[2]):
        ids, inv = np.unique(x[:, :, c], return_inverse=True)
        inv = inv.reshape(x.shape[:2])
        fg = ids != 0
        rank = np.cumsum(fg) * fg                        # 1..k for the non-zero ids in ascending order, 0 for id 0
        lab = rank[inv]
        out = np.where(lab > 0, lab + count, out)
        count += int(fg.sum())
    return out.astype("int32")


def pair_coordinates(set_a: np.ndarray, set_b: np.ndarray, radius: float):
    """Minimum-total-distance unique pairing of two point sets, pairs farther apart than `radius` discarded
    (reference `utils/tools.py:104-147`): (pairing [k, 2] of (index in A, index in B), unpaired A, unpaired B)."""
    from scipy.optimize import linear_sum_assignment
    from scipy.spatial.distance import cdist
    cost = cdist(set_a, set_b, metric="euclidean")
    ia, ib = linear_sum_assignment(cost)
    keep = cost[ia, ib] <= radius
    pa, pb = ia[keep], ib[keep]
    pairing = np.concatenate([pa[:, None], pb[:, None]], axis=-1)
    return pairing, np.delete(np.arange(set_a.shape[0]), pa), np.delete(np.arange(set_b.shape[0]), pb)


def cell_detection_scores(paired_true, paired_pred, unpaired_true, unpaired_pred, w=(1, 1)):
    """(f1, precision, recall) of the detection task from the pairing (reference `metrics.py:221-236`)."""
    tp, fp, fn = paired_pred.shape[0], unpaired_pred.shape[0], unpaired_true.shape[0]
    return 2 * tp / (2 * tp + w[0] * fp + w[1] * fn), tp / (tp + fp), tp / (tp + fn)


def cell_type_detection_scores(paired_true, paired_pred, unpaired_true, unpaired_pred, type_id, w=(2, 2, 1, 1),
                               exhaustive: bool = True):
    """(f1, precision, recall) of one nucleus type (reference `metrics.py:239-274`)."""
    sel = (paired_true == type_id) | (paired_pred == type_id)
    pt, pp = paired_true[sel], paired_pred[sel]
    tp = ((pt == type_id) & (pp == type_id)).sum()
    tn = ((pt != type_id) & (pp != type_id)).sum()
    fp = ((pt != type_id) & (pp == type_id)).sum()
    fn = ((pt == type_id) & (pp != type_id)).sum()
    if not exhaustive:
        fp -= (pt == -1).sum()
    fp_d, fn_d = (unpaired_pred == type_id).sum(), (unpaired_true == type_id).sum()
    prec = (tp + tn) / (tp + tn + w[0] * fp + w[2] * fp_d)
    rec = (tp + tn) / (tp + tn + w[1] * fn + w[3] * fn_d)
    f1 = (2 * (tp + tn)) / (2 * (tp + tn) + w[0] * fp + w[1] * fn + w[2] * fp_d + w[3] * fn_d)
    return f1, prec, rec


def binary_dice(pred: np.ndarray, target: np.ndarray) -> float:
    """torchmetrics.functional.dice(preds, target, ignore_index=0) on {0, 1} maps (caller :817-822): the micro dice of
    the foreground class, 2 tp / (2 tp + fp + fn), 0 when the denominator is 0."""
    p, t = np.asarray(pred) > 0, np.asarray(target) > 0
    tp, fp, fn = int((p & t).sum()), int((p & ~t).sum()), int((~p & t).sum())
    den = 2 * tp + fp + fn
    return 2.0 * tp / den if den else 0.0


def binary_jaccard(pred: np.ndarray, target: np.ndarray) -> float:
    """torchmetrics.functional.classification.binary_jaccard_index (caller :826-833): tp / (tp + fp + fn), 0 when empty."""
    p, t = np.asarray(pred) > 0, np.asarray(target) > 0
    tp, fp, fn = int((p & t).sum()), int((p & ~t).sum()), int((~p & t).sum())
    den = tp + fp + fn
    return float(tp) / den if den else 0.0
