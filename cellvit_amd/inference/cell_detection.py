"""WSI inference CLI on the MI355X engine — same command line as the reference
(/root/reference/cell_segmentation/inference/cell_detection.py:906-1006):

    python -m cellvit_amd.inference.cell_detection --model CKPT [--gpu 0] [--magnification 40] [--enforce_amp]
        [--batch_size 8] [--outdir_subdir NAME] [--geojson]
        process_wsi --wsi_path SLIDE --patched_slide_path DIR
      | process_dataset --wsi_paths DIR --patch_dataset_path DIR [--filelist CSV] [--wsi_extension svs]

Inputs: a reference checkpoint ``{arch, config (flattened with '.'), model_state_dict}`` (base_trainer.py:229-245)
and a pre-patched slide directory (``metadata.yaml``, ``patch_metadata.json``, ``patches/*.png``,
wsi_datamodel.py:50-146).  Outputs under ``<patched_slide_path>/cell_detection[/<subdir>]``: ``cells.json``,
``cell_detection.json``, optional ``*.geojson``, ``cells.pt`` (cell_detection.py:438-475).

Per tile everything up to the instance records runs on the GPU (forward, post-processing, token pooling); with
``torch.distributed`` initialised (one process per GPU) the tile list is sharded and margin-cell records are
all-gathered for the slide-level de-duplication (row f1 of SURVEY §8: the reference uses shapely STRtree polygon
intersections, unavailable here — this module applies the same rules with its own exact polygon-intersection area).
"""
from __future__ import annotations

import argparse
import csv
import json
import logging
import math
from collections import defaultdict
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import yaml

from .. import sharding as S
from ..model import build_model
from ..postproc import _params, postprocess_device, records_to_dicts

COLOR_DICT = {1: [255, 0, 0], 2: [34, 221, 77], 3: [35, 92, 236], 4: [254, 255, 0], 5: [255, 159, 68]}   # :76-82
TYPE_NUCLEI_DICT = {1: "Neoplastic", 2: "Inflammatory", 3: "Connective", 4: "Dead", 5: "Epithelial"}       # :84-90


def unflatten_dict(d: dict, sep: str = ".") -> dict:
    """utils/tools.py:176-194."""
    out: dict = {}
    for key, value in d.items():
        parts = key.split(sep)
        cur = out
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = value
    return out


class PatchedSlide:
    """Reader of a pre-patched slide directory (datamodel/wsi_datamodel.py:50-146)."""

    def __init__(self, name: str, patched_slide_path: str):
        self.name = name
        self.patched_slide_path = Path(patched_slide_path).resolve()
        with open(self.patched_slide_path / "metadata.yaml") as f:
            self.metadata = yaml.safe_load(f)
        self.metadata["label_map_inverse"] = {v: k for k, v in self.metadata.get("label_map", {}).items()}
        with open(self.patched_slide_path / "patch_metadata.json") as f:
            meta = json.load(f)
        self.patches_list = [str(list(e.keys())[0]) for e in meta]
        self.all_patch_metadata = {str(list(e.keys())[0]): e[str(list(e.keys())[0])] for e in meta}

    def load_patch(self, patch_name: str) -> Tuple[np.ndarray, dict]:
        from PIL import Image
        img = np.asarray(Image.open(self.patched_slide_path / "patches" / patch_name).convert("RGB"))
        md = dict(self.all_patch_metadata[patch_name])
        mp = md.get("metadata_path")
        if mp and (self.patched_slide_path / mp).exists():
            with open(self.patched_slide_path / mp) as f:
                md.update(yaml.safe_load(f) or {})
        md["name"] = patch_name
        return img, md


def check_wsi(wsi: PatchedSlide, magnification: float = 40.0) -> None:
    """cell_detection.py:1009-1039."""
    assert wsi.metadata["magnification"] == magnification, "The slide must be patched at the network magnification"
    assert wsi.metadata["patch_size"] == 1024, "The patch-size must be 1024 (for 40x)"
    assert wsi.metadata["patch_overlap"] == 64, "The patch-overlap must be 64 pixels"


class CellSegmentationInference:
    """cell_detection.py:92-242: load the checkpoint, build the model from `arch` + `config`, set up precision."""

    def __init__(self, model_path: str, gpu: int, enforce_mixed_precision: bool = False) -> None:
        self.logger = logging.getLogger("cellvit_amd")
        self.device = torch.device("cuda", gpu)
        ckpt = torch.load(str(model_path), map_location="cpu", weights_only=False)
        self.run_conf = unflatten_dict(ckpt["config"], ".")
        self.mixed_precision = bool(enforce_mixed_precision or
                                    self.run_conf.get("training", {}).get("mixed_precision", False))
        self.model = build_model(ckpt["arch"], self.run_conf, compute_dtype="fp16" if self.mixed_precision else "fp32")
        self.logger.info(self.model.load_state_dict(ckpt["model_state_dict"]))
        self.model.eval()
        norm = self.run_conf.get("transformations", {}).get("normalize", {})
        self.mean = torch.tensor(norm.get("mean", (0.5, 0.5, 0.5)), dtype=torch.float32, device=self.device)
        self.std = torch.tensor(norm.get("std", (0.5, 0.5, 0.5)), dtype=torch.float32, device=self.device)

    def _normalize(self, tiles_u8: torch.Tensor) -> torch.Tensor:
        """T.ToTensor + T.Normalize (:214-227) on the device: [B,H,W,3] u8 -> [B,3,H,W] f32."""
        x = tiles_u8.to(self.device).float() / 255.0
        return ((x - self.mean) / self.std).permute(0, 3, 1, 2).contiguous()

    # ------------------------------------------------------------------------------------------
    def process_wsi(self, wsi: PatchedSlide, subdir_name: Optional[str] = None, patch_size: int = 1024,
                    overlap: int = 64, batch_size: int = 8, geojson: bool = False) -> dict:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        nuclei_types = self.run_conf["dataset_config"]["nuclei_types"]
        magnification = int(wsi.metadata["magnification"])
        obj, ks = _params(magnification)
        outdir = Path(wsi.patched_slide_path) / "cell_detection" / (subdir_name or "")
        outdir.mkdir(exist_ok=True, parents=True)
        my_tiles = S.shard_tiles(len(wsi.patches_list), rank, world, block=batch_size)
        cells: List[dict] = []
        tokens_out: List[torch.Tensor] = []
        processed = []
        with torch.no_grad():
            for b0 in range(0, len(my_tiles), batch_size):
                names = [wsi.patches_list[i] for i in my_tiles[b0:b0 + batch_size]]
                loaded = [wsi.load_patch(n) for n in names]
                x = self._normalize(torch.from_numpy(np.stack([im for im, _ in loaded])))
                pred = self.model.forward(x, retrieve_tokens=True)
                bin_am, typ_am = self.model._last_argmax           # argmax planes written by the forward kernels
                inst, recs, n_recs, contours, n_pts = postprocess_device(bin_am, typ_am, pred["hv_map"],
                                                                         self.model.num_nuclei_classes, obj, ks)
                dicts = records_to_dicts(recs, n_recs, contours, n_pts)   # <- cells leave the device here
                tokens = pred["tokens"]
                for idx, (tile_cells, (_, md)) in enumerate(zip(dicts, loaded)):
                    row, col = int(md["row"]), int(md["col"])
                    processed.append(f"{row}_{col}")
                    xg, yg = S.global_offset(row, col, wsi.metadata["patch_size"], wsi.metadata["downsampling"], overlap)
                    off = np.array([xg, yg])
                    for cell in tile_cells.values():
                        if cell["type"] == nuclei_types["Background"]:
                            continue
                        bbox = cell["bbox"]
                        d = {
                            "bbox": (bbox + off).tolist(),
                            "centroid": (cell["centroid"] + np.flip(off)).tolist(),
                            "contour": (cell["contour"] + np.flip(off)).tolist(),
                            "type_prob": cell["type_prob"], "type": cell["type"],
                            "patch_coordinates": [row, col],
                            "cell_status": S.cell_status(bbox, patch_size, overlap),
                            "offset_global": off.tolist(),
                        }
                        if np.max(bbox) == patch_size or np.min(bbox) == 0:
                            pos = S.cell_edge_position(bbox, patch_size)
                            d["edge_position"] = True
                            d["edge_information"] = {"position": pos, "edge_patches": S.edge_patches(pos, row, col)}
                        else:
                            d["edge_position"] = False
                        cells.append(d)
                        # cell token = mean of the ViT tokens under the bbox (cell_detection.py:396-409)
                        bb = bbox / self.model.patch_size
                        r0, c0 = int(math.floor(bb[0, 0])), int(math.floor(bb[0, 1]))
                        r1, c1 = int(math.ceil(bb[1, 0])), int(math.ceil(bb[1, 1]))
                        tokens_out.append(tokens[idx, :, r0:r1, c0:c1].reshape(tokens.shape[1], -1).mean(dim=1).cpu())
        self.logger.info(f"[rank {rank}] detected cells before cleaning: {len(cells)}")
        keep = stitch_cells(cells, self.logger)
        cells = [cells[i] for i in keep]
        tokens_out = [tokens_out[i] for i in keep]
        if world > 1:
            gathered: List[Optional[list]] = [None] * world
            dist.all_gather_object(gathered, cells)          # slide-level record exchange (JSON-sized, latency-bound)
            cells_all = [c for part in gathered for c in part]
            keep2 = stitch_cells(cells_all, self.logger)     # cross-rank duplicates in the overlap margins
            cells_all = [cells_all[i] for i in keep2]
        else:
            cells_all = cells
        if rank == 0:
            meta = {"wsi_metadata": wsi.metadata, "processed_patches": processed, "type_map": nuclei_types}
            with open(outdir / "cells.json", "w") as f:
                json.dump({**meta, "cells": cells_all}, f, indent=2, default=_np_default)
            det = [{"bbox": c["bbox"], "centroid": c["centroid"], "type": c["type"]} for c in cells_all]
            with open(outdir / "cell_detection.json", "w") as f:
                json.dump({**meta, "cells": det}, f, indent=2, default=_np_default)
            if geojson:
                with open(outdir / "cells.geojson", "w") as f:
                    json.dump(convert_geojson(cells_all, True), f, indent=2, default=_np_default)
                with open(outdir / "cell_detection.geojson", "w") as f:
                    json.dump(convert_geojson(cells_all, False), f, indent=2, default=_np_default)
        if tokens_out:
            torch.save({"x": torch.stack(tokens_out),
                        "positions": torch.tensor([c["centroid"] for c in cells], dtype=torch.float32),
                        "contours": [torch.tensor(c["contour"], dtype=torch.float32) for c in cells],
                        "metadata": {"wsi_metadata": wsi.metadata, "nuclei_types": nuclei_types}},
                       outdir / (f"cells.pt" if world == 1 else f"cells_rank{rank}.pt"))
        return {"n_cells": len(cells_all), "outdir": str(outdir)}


def _np_default(o):
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, np.ndarray):
        return o.tolist()
    raise TypeError(type(o))


# ----------------------------------------------------------------------------------------------------
# slide-level de-duplication (CellPostProcessor, cell_detection.py:600-767) — same rules, exact polygon geometry
# ----------------------------------------------------------------------------------------------------
def _poly_area(contour: np.ndarray) -> float:
    """Area of the closed polygon through the contour points (shoelace), as `shapely.Polygon(contour).area`."""
    pts = np.asarray(contour, dtype=np.float64)
    if len(pts) < 3:
        return 0.0
    x, y = pts[:, 0], pts[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def _edge_crossings_y(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """y coordinates of all proper intersection points between the edges of polygons a and b."""
    a0, a1 = a, np.roll(a, -1, axis=0)
    b0, b1 = b, np.roll(b, -1, axis=0)
    da, db = (a1 - a0)[:, None, :], (b1 - b0)[None, :, :]
    w = (b0[None, :, :] - a0[:, None, :])
    den = da[..., 0] * db[..., 1] - da[..., 1] * db[..., 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (w[..., 0] * db[..., 1] - w[..., 1] * db[..., 0]) / den
        u = (w[..., 0] * da[..., 1] - w[..., 1] * da[..., 0]) / den
    ok = (den != 0) & (t > 0) & (t < 1) & (u > 0) & (u < 1)
    ys = a0[:, None, 1] + t * da[..., 1]
    return ys[ok]


def _x_intervals(poly: np.ndarray, yc: float) -> np.ndarray:
    """Sorted x coordinates where the horizontal line y = yc crosses the polygon's edges (even-odd interior:
    [x0, x1], [x2, x3], ...)."""
    p0, p1 = poly, np.roll(poly, -1, axis=0)
    y0, y1 = p0[:, 1], p1[:, 1]
    hit = ((y0 <= yc) & (yc < y1)) | ((y1 <= yc) & (yc < y0))
    xs = p0[hit, 0] + (yc - y0[hit]) * (p1[hit, 0] - p0[hit, 0]) / (y1[hit] - y0[hit])
    return np.sort(xs)


def _intersection_area(a: np.ndarray, b: np.ndarray) -> float:
    """EXACT area of the intersection of two polygons (even-odd interiors) by slab decomposition: between two
    consecutive event ordinates (vertices of either polygon, crossings of an a-edge with a b-edge) every interval end
    point is linear in y, so the common length L(y) is linear and the midpoint rule integrates it exactly."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if len(a) < 3 or len(b) < 3:
        return 0.0
    lo, hi = max(a[:, 1].min(), b[:, 1].min()), min(a[:, 1].max(), b[:, 1].max())
    if hi <= lo or max(a[:, 0].min(), b[:, 0].min()) >= min(a[:, 0].max(), b[:, 0].max()):
        return 0.0
    ev = np.concatenate([a[:, 1], b[:, 1], _edge_crossings_y(a, b), [lo, hi]])
    ev = np.unique(ev[(ev >= lo) & (ev <= hi)])
    area = 0.0
    for y0, y1 in zip(ev[:-1], ev[1:]):
        ym = 0.5 * (y0 + y1)
        xa, xb = _x_intervals(a, ym), _x_intervals(b, ym)
        length = 0.0
        for i in range(0, len(xa) - 1, 2):
            for j in range(0, len(xb) - 1, 2):
                length += max(0.0, min(xa[i + 1], xb[j + 1]) - max(xa[i], xb[j]))
        area += length * (y1 - y0)
    return area


def _overlap_fractions(ca: dict, cb: dict) -> Tuple[float, float, float, float]:
    """(intersection / area_a, intersection / area_b, area_a, area_b) of two cells' contour polygons — the quantities
    the reference takes from shapely (`cell_detection.py:722-747`), computed exactly (no shapely here)."""
    a, b = np.asarray(ca["contour"]), np.asarray(cb["contour"])
    aa, ab = _poly_area(a), _poly_area(b)
    inter = _intersection_area(a, b) if aa > 0 and ab > 0 else 0.0
    return (inter / aa if aa else 0.0), (inter / ab if ab else 0.0), aa, ab


def stitch_cells(cells: List[dict], logger: Optional[logging.Logger] = None) -> List[int]:
    """Indices of the cells to keep: mid cells; margin cells; edge cells only if the neighbouring tile (first
    `edge_patches` entry) produced no margin cells (:645-674); then up to 20 rounds of overlap removal where of every
    group of cells overlapping by > 1 % of either area the largest *other* cell survives (:676-767)."""
    idx_margin = [i for i, c in enumerate(cells) if c["cell_status"] != 0]
    keep = [i for i, c in enumerate(cells) if c["cell_status"] == 0]
    existing = {f"{cells[i]['patch_coordinates'][0]}_{cells[i]['patch_coordinates'][1]}" for i in idx_margin}
    cleaned = []
    for i in idx_margin:
        c = cells[i]
        if not c["edge_position"]:
            cleaned.append(i)
        else:
            ep = c["edge_information"]["edge_patches"]
            if ep is None or f"{ep[0][0]}_{ep[0][1]}" not in existing:
                cleaned.append(i)
    merged = sorted(cleaned)
    for iteration in range(20):
        grid: Dict[Tuple[int, int], List[int]] = defaultdict(list)
        for i in merged:
            (r0, c0), (r1, c1) = cells[i]["bbox"]
            for gy in range(int(r0) // 64, int(r1) // 64 + 1):
                for gx in range(int(c0) // 64, int(c1) // 64 + 1):
                    grid[(gy, gx)].append(i)
        done, out, overlaps = set(), [], 0
        for i in merged:
            if i in done:
                continue
            (r0, c0), (r1, c1) = cells[i]["bbox"]
            cand = set()
            for gy in range(int(r0) // 64, int(r1) // 64 + 1):
                for gx in range(int(c0) // 64, int(c1) // 64 + 1):
                    cand.update(grid[(gy, gx)])
            sub = []
            for j in sorted(cand):
                if j == i or j in done:
                    continue
                (a0, b0), (a1, b1) = cells[j]["bbox"]
                if a0 >= r1 or a1 <= r0 or b0 >= c1 or b1 <= c0:
                    continue
                fa, fb, _, area_j = _overlap_fractions(cells[i], cells[j])
                if fa > 0.01 or fb > 0.01:
                    overlaps += 1
                    sub.append((area_j, j))
                    done.add(j)
            out.append(i if not sub else max(sub)[1])
            done.add(i)
        if logger:
            logger.info(f"Iteration {iteration}: Found overlap of # cells: {overlaps}")
        merged = sorted(set(out))
        if overlaps == 0:
            break
    return sorted(keep + merged)


def convert_geojson(cell_list: List[dict], polygons: bool = False) -> List[dict]:
    """cell_detection.py:538-597 + template_geojson.py:9-52: one MultiPolygon / MultiPoint feature per cell type."""
    by_type: Dict[int, list] = defaultdict(list)
    for c in cell_list:
        if polygons:
            ring = [list(map(float, p)) for p in c["contour"]]
            ring.append(ring[0])
            by_type[c["type"]].append([ring])
        else:
            by_type[c["type"]].append([float(c["centroid"][0]), float(c["centroid"][1])])
    feats = []
    for t, geoms in by_type.items():
        feats.append({
            "type": "Feature", "id": f"cellvit_amd-{t}",
            "geometry": {"type": "MultiPolygon" if polygons else "MultiPoint", "coordinates": geoms},
            "properties": {"objectType": "annotation",
                           "classification": {"name": TYPE_NUCLEI_DICT.get(t, str(t)), "color": COLOR_DICT.get(t, [0, 0, 0])}},
        })
    return feats


class InferenceWSIParser:
    """cell_detection.py:906-1006 — identical flags."""

    def __init__(self) -> None:
        p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                    description="Perform CellViT inference for given run-directory with model checkpoints")
        req = p.add_argument_group("required named arguments")
        req.add_argument("--model", type=str, required=True, help="Model checkpoint file that is used for inference")
        p.add_argument("--gpu", type=int, default=0, help="Cuda-GPU ID for inference")
        p.add_argument("--magnification", type=float, default=40, help="Network magnification")
        p.add_argument("--enforce_amp", action="store_true", help="Use mixed precision for inference (enforced)")
        p.add_argument("--batch_size", type=int, default=8, help="Inference batch-size")
        p.add_argument("--outdir_subdir", type=str, default=None)
        p.add_argument("--geojson", action="store_true")
        sub = p.add_subparsers(dest="command", description="process_wsi | process_dataset")
        w = sub.add_parser("process_wsi", description="Process a single WSI file")
        w.add_argument("--wsi_path", type=str)
        w.add_argument("--patched_slide_path", type=str)
        d = sub.add_parser("process_dataset", description="Process a whole dataset")
        d.add_argument("--wsi_paths", type=str)
        d.add_argument("--patch_dataset_path", type=str)
        d.add_argument("--filelist", type=str, default=None)
        d.add_argument("--wsi_extension", type=str, default="svs")
        self.parser = p

    def parse_arguments(self, argv=None) -> dict:
        return vars(self.parser.parse_args(argv))


def main(argv=None) -> None:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)s] %(message)s")
    conf = InferenceWSIParser().parse_arguments(argv)
    import os
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
        conf["gpu"] = int(os.environ.get("LOCAL_RANK", conf["gpu"]))
    torch.cuda.set_device(conf["gpu"])
    inf = CellSegmentationInference(conf["model"], conf["gpu"], conf["enforce_amp"])
    if conf["command"].lower() == "process_wsi":
        slide = PatchedSlide(Path(conf["wsi_path"]).stem, conf["patched_slide_path"])
        check_wsi(slide, conf["magnification"])
        inf.process_wsi(slide, conf["outdir_subdir"], batch_size=conf["batch_size"], geojson=conf["geojson"])
    elif conf["command"].lower() == "process_dataset":
        if conf["filelist"]:
            with open(conf["filelist"]) as f:
                names = [r["Filename"] for r in csv.DictReader(f)]
        else:
            names = [p.name for p in sorted(Path(conf["wsi_paths"]).glob(f"**/*.{conf['wsi_extension']}"))]
        for n in names:
            pdir = Path(conf["patch_dataset_path"]) / Path(n).stem
            if not (pdir / "metadata.yaml").exists():
                logging.warning(f"slide {n} is not patched under {pdir} — skipped")
                continue
            slide = PatchedSlide(Path(n).stem, str(pdir))
            check_wsi(slide, conf["magnification"])
            inf.process_wsi(slide, conf["outdir_subdir"], batch_size=conf["batch_size"], geojson=conf["geojson"])
    else:
        raise ValueError("Unknown command")


if __name__ == "__main__":
    main()
