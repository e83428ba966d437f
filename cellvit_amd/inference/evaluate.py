"""Patch-wise evaluation caller — the PanNuke-style experiment inference of the reference
(`cell_segmentation/inference/inference_cellvit_experiment_pannuke.py:599-996`) around the MI355X hot path.

Same stages and the same result dictionaries as the reference class:

  inference_step        :599-651   forward (fp16 engine under `mixed_precision`) -> unpack -> step metrics
  unpack_predictions    :653-701   softmax, `model.calculate_instance_map`, `model.generate_instance_nuclei_map`
  unpack_masks          :703-746   one-hot ground truth, per-class instance maps, `calculate_instances` (device records)
  calculate_step_metric :748-963   binary dice / jaccard, bPQ (`binarize` + PQ), per-class PQ, centroid pairing
  run                   :340-597   dataset / tissue / nucleus-type aggregation -> the `inference_results.json` dict

The network, the post-processing of the predictions and the instance records of the ground truth run on the GPU through
the C-ABI; the metrics are host numpy on the record arrays and label maps, as in the reference.  Plotting and logging
are not part of this path.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from ..metrics import (binarize, binary_dice, binary_jaccard, cell_detection_scores, cell_type_detection_scores,
                       pair_coordinates, panoptic_quality, remap_label)
from ..postproc import calculate_instances


class PatchEvaluator:
    """`InferenceCellViT` of the reference reduced to its inference path (the experiment-directory / checkpoint
    bookkeeping of :60-338 belongs to the training framework and is out of scope)."""

    def __init__(self, model, dataset_config: dict, num_classes: Optional[int] = None, magnification: int = 40,
                 mixed_precision: bool = False, device: Optional[torch.device] = None):
        self.model = model
        self.dataset_config = dataset_config            # {"tissue_types": {name: idx}, "nuclei_types": {name: idx}}
        self.num_classes = int(num_classes if num_classes is not None else len(dataset_config["nuclei_types"]))
        self.magnification = magnification
        self.mixed_precision = mixed_precision
        self.device = device or torch.device("cuda", torch.cuda.current_device())

    # ------------------------------------------------------------------ :599-651
    def inference_step(self, batch: tuple) -> dict:
        imgs = batch[0].to(self.device)
        masks, tissue_types, image_names = batch[1], list(batch[2]), list(batch[3])
        if self.mixed_precision:
            with torch.autocast(device_type="cuda", dtype=torch.float16):
                predictions = self.model.forward(imgs)
        else:
            predictions = self.model.forward(imgs)
        predictions = self.unpack_predictions(predictions)
        gt = self.unpack_masks(masks, tissue_types)
        batch_metrics, _ = self.calculate_step_metric(predictions, gt, image_names)
        batch_metrics["tissue_types"] = tissue_types
        return batch_metrics

    # ------------------------------------------------------------------ :653-701
    def unpack_predictions(self, predictions: dict) -> dict:
        predictions = dict(predictions)
        predictions["tissue_types"] = predictions["tissue_types"].to(self.device)
        predictions["nuclei_binary_map"] = F.softmax(predictions["nuclei_binary_map"], dim=1)
        predictions["nuclei_type_map"] = F.softmax(predictions["nuclei_type_map"], dim=1)
        predictions["instance_map"], predictions["instance_types"] = self.model.calculate_instance_map(
            predictions, magnification=self.magnification)
        predictions["instance_types_nuclei"] = self.model.generate_instance_nuclei_map(
            predictions["instance_map"], predictions["instance_types"]).to(self.device)
        predictions["batch_size"] = predictions["tissue_types"].shape[0]
        return predictions

    # ------------------------------------------------------------------ :703-746
    def unpack_masks(self, masks: dict, tissue_types: List[str]) -> dict:
        nb = F.one_hot(masks["nuclei_binary_map"].long(), num_classes=2).float()
        nt = F.one_hot(masks["nuclei_type_map"].long().reshape(masks["instance_map"].shape), num_classes=self.num_classes).float()
        gt = {
            "nuclei_type_map": nt.permute(0, 3, 1, 2).to(self.device),
            "nuclei_binary_map": nb.permute(0, 3, 1, 2).to(self.device),
            "hv_map": masks["hv_map"].to(self.device),
            "instance_map": masks["instance_map"].to(self.device),
            "instance_types_nuclei": (nt * masks["instance_map"][..., None]).permute(0, 3, 1, 2).to(self.device),
            "tissue_types": torch.tensor([self.dataset_config["tissue_types"][t] for t in tissue_types],
                                         dtype=torch.long, device=self.device),
        }
        gt["instance_types"] = calculate_instances(gt["nuclei_type_map"], gt["instance_map"])
        gt["batch_size"] = gt["tissue_types"].shape[0]
        return gt

    # ------------------------------------------------------------------ :748-963
    def calculate_step_metric(self, predictions: dict, gt: dict, image_names: List[str]) -> Tuple[dict, list]:
        pred_tissue = torch.argmax(F.softmax(predictions["tissue_types"], dim=-1), dim=-1).cpu().numpy().astype(np.uint8)
        pred_inst_nuc = predictions["instance_types_nuclei"].cpu().numpy().astype("int32")
        pred_binary = torch.argmax(predictions["nuclei_binary_map"], dim=1).cpu().numpy()
        inst_gt = gt["instance_map"].cpu().numpy()
        gt_tissue = gt["tissue_types"].cpu().numpy().astype(np.uint8)
        gt_binary = torch.argmax(gt["nuclei_binary_map"], dim=1).to(torch.uint8).cpu().numpy()
        gt_inst_nuc = gt["instance_types_nuclei"].cpu().numpy().astype("int32")

        m: Dict[str, list] = {k: [] for k in ("binary_dice_scores", "binary_jaccard_scores", "pq_scores", "dq_scores", "sq_scores",
                                               "cell_type_pq_scores", "cell_type_dq_scores", "cell_type_sq_scores")}
        scores = []
        paired_all, unpaired_true_all, unpaired_pred_all, true_type_all, pred_type_all = [], [], [], [], []
        true_off = pred_off = 0
        radius = 12 if self.magnification == 40 else 6
        for i in range(len(pred_tissue)):
            dice = binary_dice(pred_binary[i], gt_binary[i])
            jac = binary_jaccard(pred_binary[i], gt_binary[i])
            m["binary_dice_scores"].append(float(dice))
            m["binary_jaccard_scores"].append(float(jac))
            if len(np.unique(inst_gt[i])) == 1:
                dq = sq = pq = np.nan
            else:
                (dq, sq, pq), _ = panoptic_quality(remap_label(inst_gt[i]), binarize(pred_inst_nuc[i][1:].transpose(1, 2, 0)))
            m["pq_scores"].append(pq); m["dq_scores"].append(dq); m["sq_scores"].append(sq)
            scores.append([dice, jac, pq])

            t_pq, t_dq, t_sq = [], [], []
            for j in range(self.num_classes):
                p_cls, g_cls = remap_label(pred_inst_nuc[i][j]), remap_label(gt_inst_nuc[i][j])
                if len(np.unique(g_cls)) == 1:          # class absent from the ground truth: skipped from the mean
                    d_ = s_ = p_ = np.nan
                else:
                    (d_, s_, p_), _ = panoptic_quality(p_cls, g_cls, match_iou=0.5)   # argument order as the caller (:882-886)
                t_pq.append(p_); t_dq.append(d_); t_sq.append(s_)
            m["cell_type_pq_scores"].append(t_pq); m["cell_type_dq_scores"].append(t_dq); m["cell_type_sq_scores"].append(t_sq)

            tc = np.array([v["centroid"] for v in gt["instance_types"][i].values()])
            tt = np.array([v["type"] for v in gt["instance_types"][i].values()])
            pc = np.array([v["centroid"] for v in predictions["instance_types"][i].values()])
            pt = np.array([v["type"] for v in predictions["instance_types"][i].values()])
            if tc.shape[0] == 0:
                tc, tt = np.array([[0, 0]]), np.array([0])
            if pc.shape[0] == 0:
                pc, pt = np.array([[0, 0]]), np.array([0])
            paired, un_t, un_p = pair_coordinates(tc, pc, radius)
            true_off = true_off + true_type_all[-1].shape[0] if i != 0 else 0
            pred_off = pred_off + pred_type_all[-1].shape[0] if i != 0 else 0
            true_type_all.append(tt); pred_type_all.append(pt)
            if paired.shape[0] != 0:
                paired = paired + np.array([[true_off, pred_off]])
                paired_all.append(paired)
            unpaired_true_all.append(un_t + true_off)
            unpaired_pred_all.append(un_p + pred_off)

        batch_metrics = dict(m)
        batch_metrics.update({
            "image_names": image_names,
            "tissue_pred": pred_tissue,
            "tissue_gt": gt_tissue,
            "paired_all": np.concatenate(paired_all, axis=0) if paired_all else np.zeros((0, 2), np.int64),
            "unpaired_true_all": np.concatenate(unpaired_true_all, axis=0),
            "unpaired_pred_all": np.concatenate(unpaired_pred_all, axis=0),
            "true_inst_type_all": np.concatenate(true_type_all, axis=0),
            "pred_inst_type_all": np.concatenate(pred_type_all, axis=0),
        })
        return batch_metrics, scores

    # ------------------------------------------------------------------ :340-597
    def run(self, batches: Iterable[tuple], outdir: Optional[Path] = None) -> dict:
        """Loop over (imgs, masks, tissue_types, image_names) batches and aggregate as `run_inference` does; with `outdir`
        the result is also written to `inference_results.json`."""
        acc: Dict[str, list] = {k: [] for k in ("image_names", "binary_dice_scores", "binary_jaccard_scores", "pq_scores", "dq_scores",
                                                 "sq_scores", "cell_type_pq_scores", "cell_type_dq_scores", "cell_type_sq_scores",
                                                 "tissue_types")}
        tissue_pred, tissue_gt = [], []
        paired_g, un_t_g, un_p_g, true_g, pred_g = [], [], [], [], []
        true_off = pred_off = 0
        for bi, batch in enumerate(batches):
            bm = self.inference_step(batch)
            for k in acc:
                acc[k] = acc[k] + list(bm[k])
            tissue_pred.append(bm["tissue_pred"]); tissue_gt.append(bm["tissue_gt"])
            true_off = true_off + true_g[-1].shape[0] if bi != 0 else 0
            pred_off = pred_off + pred_g[-1].shape[0] if bi != 0 else 0
            true_g.append(bm["true_inst_type_all"]); pred_g.append(bm["pred_inst_type_all"])
            paired_g.append(bm["paired_all"] + np.array([[true_off, pred_off]]))
            un_t_g.append(bm["unpaired_true_all"] + true_off)
            un_p_g.append(bm["unpaired_pred_all"] + pred_off)
        return self.aggregate(acc, tissue_pred, tissue_gt, paired_g, un_t_g, un_p_g, true_g, pred_g, outdir)

    def aggregate(self, acc, tissue_pred, tissue_gt, paired_g, un_t_g, un_p_g, true_g, pred_g, outdir=None) -> dict:
        tissue_inf = [t.lower() for t in acc["tissue_types"]]
        paired = np.concatenate(paired_g, axis=0).astype(np.int64)
        un_t, un_p = np.concatenate(un_t_g, axis=0).astype(np.int64), np.concatenate(un_p_g, axis=0).astype(np.int64)
        true_type, pred_type = np.concatenate(true_g, axis=0), np.concatenate(pred_g, axis=0)
        paired_true_type, paired_pred_type = true_type[paired[:, 0]], pred_type[paired[:, 1]]
        unpaired_true_type, unpaired_pred_type = true_type[un_t], pred_type[un_p]

        dice, jac = np.array(acc["binary_dice_scores"]), np.array(acc["binary_jaccard_scores"])
        pq, dq, sq = np.array(acc["pq_scores"]), np.array(acc["dq_scores"]), np.array(acc["sq_scores"])
        ct_pq, ct_dq, ct_sq = acc["cell_type_pq_scores"], acc["cell_type_dq_scores"], acc["cell_type_sq_scores"]
        tissue_acc = float(np.mean(np.concatenate(tissue_gt) == np.concatenate(tissue_pred)))     # sklearn accuracy_score
        f1_d, prec_d, rec_d = cell_detection_scores(paired_true_type, paired_pred_type, unpaired_true_type, unpaired_pred_type)
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", category=RuntimeWarning)      # nanmean of all-NaN rows, as in the reference
                dataset_metrics = {
                    "Binary-Cell-Dice-Mean": float(np.nanmean(dice)),
                    "Binary-Cell-Jacard-Mean": float(np.nanmean(jac)),
                    "Tissue-Multiclass-Accuracy": tissue_acc,
                    "bPQ": float(np.nanmean(pq)), "bDQ": float(np.nanmean(dq)), "bSQ": float(np.nanmean(sq)),
                    "mPQ": float(np.nanmean([np.nanmean(v) for v in ct_pq])),
                    "mDQ": float(np.nanmean([np.nanmean(v) for v in ct_dq])),
                    "mSQ": float(np.nanmean([np.nanmean(v) for v in ct_sq])),
                    "f1_detection": float(f1_d), "precision_detection": float(prec_d), "recall_detection": float(rec_d),
                }
                tissue_metrics = {}
                for tissue in self.dataset_config["tissue_types"].keys():
                    tissue = tissue.lower()
                    ids = np.where(np.asarray(tissue_inf) == tissue)
                    tissue_metrics[tissue] = {
                        "Dice": float(np.nanmean(dice[ids])), "Jaccard": float(np.nanmean(jac[ids])),
                        "mPQ": float(np.nanmean([np.nanmean(v) for v in np.array(ct_pq)[ids]])),
                        "bPQ": float(np.nanmean(pq[ids])),
                    }
                nuclei_pq, nuclei_dq, nuclei_sq, nuclei_d = {}, {}, {}, {}
                for name, idx in self.dataset_config["nuclei_types"].items():
                    if name.lower() == "background":
                        continue
                    nuclei_pq[name] = float(np.nanmean([v[idx] for v in ct_pq]))
                    nuclei_dq[name] = float(np.nanmean([v[idx] for v in ct_dq]))
                    nuclei_sq[name] = float(np.nanmean([v[idx] for v in ct_sq]))
                    f1_c, prec_c, rec_c = cell_type_detection_scores(paired_true_type, paired_pred_type, unpaired_true_type,
                                                                     unpaired_pred_type, idx)
                    nuclei_d[name] = {"f1_cell": float(f1_c), "prec_cell": float(prec_c), "rec_cell": float(rec_c)}
        image_metrics = {n: {"Dice": float(dice[i]), "Jaccard": float(jac[i]), "bPQ": float(pq[i])}
                         for i, n in enumerate(acc["image_names"])}
        all_metrics = {"dataset": dataset_metrics, "tissue_metrics": tissue_metrics, "image_metrics": image_metrics,
                       "nuclei_metrics_pq": nuclei_pq, "nuclei_metrics_dq": nuclei_dq, "nuclei_metrics_sq": nuclei_sq,
                       "nuclei_metrics_d": nuclei_d}
        if outdir is not None:
            outdir = Path(outdir)
            outdir.mkdir(parents=True, exist_ok=True)
            with open(outdir / "inference_results.json", "w") as f:
                json.dump(all_metrics, f, indent=2)
        return all_metrics
