"""Whole-image evaluation caller — the MoNuSeg experiment inference of the reference
(`cell_segmentation/inference/inference_cellvit_experiment_monuseg.py:210-781`, dataset side
`cell_segmentation/datasets/monuseg.py:69-110`) around the MI355X hot path.

Three routes, as the reference:

  * whole image                     forward on the full image, post-processing of the full maps            (:347-353)
  * 256-px patches, no overlap      forward on the (i j) patches, maps re-assembled, ONE post-processing    (:548-596)
  * 256-px patches, 64-px overlap   per-patch instance records, slide-style de-duplication of the margin   (:598-673)
                                    cells (`CellPostProcessor`, patch size 256, margin 64, offsets
                                    i * 256 - i * overlap), kept cells painted into one instance map        (:675-781)

The network, the post-processing and the instance records of the ground truth run on the GPU through the C-ABI; the
de-duplication is `cellvit_amd.inference.stitch` (device geometry + the library's host rounds); metrics are host numpy.
Unpinned third-party semantics (no cv2 / shapely in this environment): `cv2.fillPoly` is restated as even-odd interior at
pixel centres plus the 8-connected boundary lines between consecutive vertices.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .. import sharding as S
from ..metrics import (binary_dice, binary_jaccard, cell_detection_scores, pair_coordinates, panoptic_quality, remap_label)
from ..postproc import calculate_instances
from .stitch import stitch_margin_records

PATCH = 256


def decompose(img: torch.Tensor, patching: bool, overlap: int) -> torch.Tensor:
    """Dataset-side patching of one image [3, H, W] (`monuseg.py:91-96`) followed by the caller's re-arrangement
    (`inference_cellvit_experiment_monuseg.py:315-317`): [(i j), 3, 256, 256] with patching, [1, 3, H, W] without."""
    if not patching:
        return img[None]
    if overlap == 0:
        c, H, W = img.shape
        x = img.reshape(c, H // PATCH, PATCH, W // PATCH, PATCH).permute(0, 1, 3, 2, 4)      # "c (h i) (w j) -> c h w i j"
    else:
        x = img.unfold(1, PATCH, PATCH - overlap).unfold(2, PATCH, PATCH - overlap)            # [c, i, j, 256, 256]
    c, ni, nj = x.shape[:3]
    return x.permute(1, 2, 0, 3, 4).reshape(ni * nj, c, PATCH, PATCH)                           # "c i j w h -> (i j) c w h"


def _line_pixels(x0: int, y0: int, x1: int, y1: int):
    """8-connected line from (x0, y0) to (x1, y1), both ends included (Bresenham)."""
    dx, dy = abs(x1 - x0), -abs(y1 - y0)
    sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
    err = dx + dy
    xs, ys = [], []
    while True:
        xs.append(x0); ys.append(y0)
        if x0 == x1 and y0 == y1:
            break
        e2 = 2 * err
        if e2 >= dy:
            err += dy; x0 += sx
        if e2 <= dx:
            err += dx; y0 += sy
    return np.asarray(xs), np.asarray(ys)


def fill_poly(canvas: np.ndarray, contour: np.ndarray, value: int) -> None:
    """`cv2.fillPoly(canvas, contour[None], value)` restated: pixels whose centre lies inside the polygon (even-odd) plus the
    boundary lines between consecutive vertices; clipped to the canvas.  (cv2 is not installable here: unpinned.)"""
    pts = np.asarray(contour, dtype=np.int64).reshape(-1, 2)
    if len(pts) == 0:
        return
    H, W = canvas.shape
    x, y = pts[:, 0], pts[:, 1]
    y_lo, y_hi = max(int(y.min()), 0), min(int(y.max()), H - 1)
    p0, p1 = pts, np.roll(pts, -1, axis=0)
    for yy in range(y_lo, y_hi + 1):
        ya, yb = p0[:, 1], p1[:, 1]
        hit = ((ya <= yy) & (yy < yb)) | ((yb <= yy) & (yy < ya))
        if not hit.any():
            continue
        xs = np.sort(p0[hit, 0] + (yy - ya[hit]) * (p1[hit, 0] - p0[hit, 0]) / (yb[hit] - ya[hit]))
        for k in range(0, len(xs) - 1, 2):
            a, b = int(np.ceil(xs[k])), int(np.floor(xs[k + 1]))
            if b >= a:
                canvas[yy, max(a, 0):min(b, W - 1) + 1] = value
    for (xa, ya), (xb, yb) in zip(p0, p1):
        lx, ly = _line_pixels(int(xa), int(ya), int(xb), int(yb))
        ok = (lx >= 0) & (lx < W) & (ly >= 0) & (ly < H)
        canvas[ly[ok], lx[ok]] = value


class MoNuSegEvaluator:
    """`MoNuSegInference` of the reference reduced to its inference path (run-directory / checkpoint / plotting bookkeeping
    of :60-208 and :783-990 belongs to the experiment framework and is out of scope)."""

    def __init__(self, model, magnification: int = 40, patching: bool = False, overlap: int = 0,
                 mixed_precision: bool = False, device: Optional[torch.device] = None):
        if overlap and not patching:
            raise ValueError("overlap needs patching (inference_cellvit_experiment_monuseg.py:118-126)")
        self.model = model
        self.magnification = magnification
        self.patching, self.overlap = patching, overlap
        self.mixed_precision = mixed_precision
        self.device = device or torch.device("cuda", torch.cuda.current_device())

    # ------------------------------------------------------------------ :300-353
    def inference_step(self, img: torch.Tensor, mask: dict, image_name: str) -> dict:
        """img: [1 | (i j), 3, h, w] as `decompose` returns it; mask: {"instance_map" [1,H,W], "nuclei_binary_map" [1,H,W]}."""
        img = img.to(self.device)
        mask = dict(mask)
        mask["instance_types"] = calculate_instances(torch.unsqueeze(mask["nuclei_binary_map"], dim=0).to(self.device),
                                                     mask["instance_map"].to(self.device))
        if self.mixed_precision:
            with torch.autocast(device_type="cuda", dtype=torch.float16):
                predictions_ = self.model.forward(img)
        else:
            predictions_ = self.model.forward(img)
        if self.overlap == 0:
            if self.patching:
                predictions_ = self.post_process_patching(predictions_)
            predictions = self.get_cell_predictions(predictions_)
            return self.calculate_step_metric(predictions, mask, [image_name])
        cell_list = self.post_process_patching_overlap(predictions_, self.overlap)
        return self.calculate_step_metric_overlap(cell_list, mask, [image_name])[0]

    # ------------------------------------------------------------------ :472-546
    @staticmethod
    def convert_binary_type(instance_types: dict) -> dict:
        out = {}
        for key, elem in instance_types.items():
            if elem["type"] == 0:
                continue
            elem["type"] = 0
            out[key] = elem
        return out

    def get_cell_predictions(self, predictions: dict) -> dict:
        predictions = dict(predictions)
        predictions["nuclei_binary_map"] = F.softmax(predictions["nuclei_binary_map"], dim=1)
        predictions["nuclei_type_map"] = F.softmax(predictions["nuclei_type_map"], dim=1)
        predictions["instance_map"], types = self.model.calculate_instance_map(predictions, magnification=self.magnification)
        predictions["instance_types"] = self.convert_binary_type(types[0])
        return predictions

    # ------------------------------------------------------------------ :548-596
    @staticmethod
    def post_process_patching(predictions: dict) -> dict:
        predictions = dict(predictions)
        n = int(np.sqrt(predictions["nuclei_binary_map"].shape[0]))
        for k in ("nuclei_binary_map", "hv_map", "nuclei_type_map"):
            v = predictions[k]                                           # "(i j) d w h -> d (i w) (j h)"
            d, w, h = v.shape[1:]
            predictions[k] = v.reshape(n, n, d, w, h).permute(2, 0, 3, 1, 4).reshape(d, n * w, n * h)[None]
        return predictions

    # ------------------------------------------------------------------ :598-673
    def post_process_patching_overlap(self, predictions: dict, overlap: int) -> List[dict]:
        predictions = dict(predictions)
        predictions["nuclei_binary_map"] = F.softmax(predictions["nuclei_binary_map"], dim=1)
        predictions["nuclei_type_map"] = F.softmax(predictions["nuclei_type_map"], dim=1)
        predictions["instance_map"], predictions["instance_types"] = self.model.calculate_instance_map(
            predictions, magnification=self.magnification)
        return self.merge_predictions(predictions, overlap)

    def merge_predictions(self, predictions: dict, overlap: int) -> List[dict]:
        """Per-patch cells -> image coordinates -> `CellPostProcessor` (patch 256, margin 64).  The cells travel as packed
        record arrays through `stitch_margin_records`; dicts are built once, for the cells that survive."""
        n = int(np.sqrt(predictions["nuclei_binary_map"].shape[0]))
        irs, frs, cts, xgs, ygs = [], [], [], [], []
        for i in range(n):
            for j in range(n):
                xg, yg = i * PATCH - i * overlap, j * PATCH - j * overlap
                for cid, cell in predictions["instance_types"][i * n + j].items():
                    if cell["type"] == 0:
                        continue
                    bb = np.asarray(cell["bbox"])
                    cont = np.asarray(cell["contour"], np.int32).reshape(-1, 2)
                    irs.append([i, j, bb[0, 0], bb[0, 1], bb[1, 0], bb[1, 1], cell["type"], S.cell_status(bb, PATCH, 64),
                                int(np.max(bb) == PATCH or np.min(bb) == 0), i * n + j, len(cont), cid])
                    frs.append([cell["centroid"][0], cell["centroid"][1], cell["type_prob"]])
                    cts.append(cont); xgs.append(xg); ygs.append(yg)
        if not irs:
            return []
        ir = np.asarray(irs, np.int32).reshape(-1, S.N_ICOL)
        fr = np.asarray(frs, np.float64).reshape(-1, S.N_FCOL)
        xg, yg = np.asarray(xgs, np.int64), np.asarray(ygs, np.int64)
        lens = ir[:, S.I_CLEN].astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ct = np.concatenate(cts).astype(np.int32)
        is_margin = ir[:, S.I_STATUS] != 0
        m_idx = np.nonzero(is_margin)[0]
        m_ct = np.concatenate([ct[offs[k]:offs[k + 1]] for k in m_idx]) if len(m_idx) else np.zeros((0, 2), np.int32)
        keep_m = stitch_margin_records(ir[m_idx], m_ct, PATCH, 1, overlap, device=self.device, offsets=(xg[m_idx], yg[m_idx]))
        keep = np.sort(np.concatenate([np.nonzero(~is_margin)[0], m_idx[keep_m]]))
        out = []
        for k in keep:
            off = np.array([xg[k], yg[k]])
            bb = ir[k, S.I_RMIN:S.I_CMAX + 1].reshape(2, 2)
            d = {"bbox": (bb + off).tolist(), "centroid": (fr[k, :2] + np.flip(off)).tolist(),
                 "contour": (ct[offs[k]:offs[k + 1]] + np.flip(off)).tolist(), "type_prob": float(fr[k, 2]), "type": int(ir[k, S.I_TYPE]),
                 "patch_coordinates": [int(ir[k, S.I_ROW]), int(ir[k, S.I_COL])], "cell_status": int(ir[k, S.I_STATUS]),
                 "offset_global": off.tolist()}
            if ir[k, S.I_EDGE]:
                pos = S.cell_edge_position(bb, PATCH)
                d["edge_position"] = True
                d["edge_information"] = {"position": pos, "edge_patches": S.edge_patches(pos, int(ir[k, S.I_ROW]), int(ir[k, S.I_COL]))}
            else:
                d["edge_position"] = False
            out.append(d)
        return out

    # ------------------------------------------------------------------ :387-470
    def _detection(self, true_centroids: np.ndarray, pred_centroids: np.ndarray):
        if true_centroids.shape[0] == 0:
            true_centroids = np.array([[0, 0]])
        if pred_centroids.shape[0] == 0:
            pred_centroids = np.array([[0, 0]])
        radius = 12 if self.magnification == 40 else 6
        paired, un_t, un_p = pair_coordinates(true_centroids, pred_centroids, radius)
        return cell_detection_scores(paired_true=paired[:, 0], paired_pred=paired[:, 1], unpaired_true=un_t, unpaired_pred=un_p)

    def calculate_step_metric(self, predictions: dict, gt: dict, image_name: List[str]) -> dict:
        inst_pred = predictions["instance_map"].detach().cpu().numpy()
        inst_gt = gt["instance_map"].detach().cpu().numpy()
        pred_bin = torch.argmax(predictions["nuclei_binary_map"], dim=1).cpu().numpy()
        tgt_bin = gt["nuclei_binary_map"].cpu().numpy()
        (dq, sq, pq), _ = panoptic_quality(remap_label(inst_gt), remap_label(inst_pred))
        f1_d, prec_d, rec_d = self._detection(np.array([v["centroid"] for v in gt["instance_types"][0].values()]),
                                              np.array([v["centroid"] for v in predictions["instance_types"].values()]))
        return {"image_name": image_name, "binary_dice_score": binary_dice(pred_bin, tgt_bin),
                "binary_jaccard_score": binary_jaccard(pred_bin, tgt_bin), "pq_score": pq, "dq_score": dq, "sq_score": sq,
                "f1_d": f1_d, "prec_d": prec_d, "rec_d": rec_d}

    # ------------------------------------------------------------------ :675-781
    def calculate_step_metric_overlap(self, cell_list: List[dict], gt: dict, image_name: List[str]) -> Tuple[dict, dict]:
        h, w = gt["nuclei_binary_map"].shape[1:]
        inst = np.zeros((h, w), dtype=np.int32)
        for instance, cell in enumerate(cell_list):        # (the reference paints cell 0 with label 0: it stays background, :707-709)
            fill_poly(inst, np.array(cell["contour"]), instance)
        pred_arr = np.clip(inst, 0, 1)
        tgt_bin = gt["nuclei_binary_map"].cpu().numpy().squeeze()
        inst_gt = gt["instance_map"].detach().cpu().numpy()
        (dq, sq, pq), _ = panoptic_quality(remap_label(inst_gt), remap_label(inst)[None])
        f1_d, prec_d, rec_d = self._detection(np.array([v["centroid"] for v in gt["instance_types"][0].values()]),
                                              np.array([v["centroid"] for v in cell_list]))
        metrics = {"image_name": image_name, "binary_dice_score": binary_dice(pred_arr, tgt_bin),
                   "binary_jaccard_score": binary_jaccard(pred_arr, tgt_bin), "pq_score": pq, "dq_score": dq, "sq_score": sq,
                   "f1_d": f1_d, "prec_d": prec_d, "rec_d": rec_d}
        types = {k + 1: dict(v, contour=np.array(v["contour"])) for k, v in enumerate(cell_list)}
        predictions = {"instance_map": torch.Tensor(inst)[None], "instance_types": types,
                       "nuclei_binary_map": F.one_hot(torch.from_numpy(pred_arr).long(), num_classes=2).permute(2, 0, 1)[None]}
        return metrics, predictions

    # ------------------------------------------------------------------ :249-297
    @staticmethod
    def aggregate(image_metrics: List[dict]) -> Dict[str, float]:
        g = lambda k: np.array([float(m[k]) for m in image_metrics])   # noqa: E731
        return {"Binary-Cell-Dice-Mean": float(np.nanmean(g("binary_dice_score"))),
                "Binary-Cell-Jacard-Mean": float(np.nanmean(g("binary_jaccard_score"))),
                "bPQ": float(np.nanmean(g("pq_score"))), "bDQ": float(np.nanmean(g("dq_score"))), "bSQ": float(np.nanmean(g("sq_score"))),
                "f1_detection": float(np.nanmean(g("f1_d"))), "precision_detection": float(np.nanmean(g("prec_d"))),
                "recall_detection": float(np.nanmean(g("rec_d")))}

    def run(self, dataset) -> Tuple[Dict[str, float], List[dict]]:
        """dataset: iterable of (image [3,H,W] float in [0,1] normalised as the model expects, mask dict, name)."""
        per_image = []
        for img, mask, name in dataset:
            masks = {k: (v[None] if v.dim() == 2 else v) for k, v in mask.items() if k in ("instance_map", "nuclei_binary_map")}
            per_image.append(self.inference_step(decompose(img, self.patching, self.overlap), masks, name))
        return self.aggregate(per_image), per_image
