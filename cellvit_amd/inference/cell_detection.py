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
from .._lib import REC_DTYPE
from ..postproc import _params, check_capacity, pool_cell_tokens, postprocess_device

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

    def load_patch_image(self, patch_name: str) -> np.ndarray:
        """Decoded RGB tile as uint8 [H, W, 3] (wsi_datamodel.py:121-136; the transform runs on the GPU here)."""
        from PIL import Image
        return np.asarray(Image.open(self.patched_slide_path / "patches" / patch_name).convert("RGB"))

    def load_patch_metadata(self, patch_name: str) -> dict:
        """wsi_datamodel.py:88-107: json entry (row, col, ...) merged with the per-patch yaml, plus `name`."""
        md = dict(self.all_patch_metadata[patch_name])
        mp = md.get("metadata_path")
        if mp and (self.patched_slide_path / mp).exists():
            with open(self.patched_slide_path / mp) as f:
                md.update(yaml.safe_load(f) or {})
        md["name"] = patch_name
        return md

    def load_patch(self, patch_name: str) -> Tuple[np.ndarray, dict]:
        return self.load_patch_image(patch_name), self.load_patch_metadata(patch_name)


class TilePrefetcher:
    """Input side of the tile loop (reference: DataLoader with ¾·cpu workers, cell_detection.py:266-282): decode worker
    threads (PIL releases the GIL while it inflates a PNG) fill PINNED uint8 batches, `depth` batches ahead of the GPU;
    the batch is handed to the device with one asynchronous 3 MB-per-tile copy.  Normalisation does not happen here —
    the forward kernels read the raw bytes (cv_forward_u8)."""

    def __init__(self, wsi: PatchedSlide, tile_ids: List[int], batch_size: int, device: torch.device,
                 num_workers: Optional[int] = None, depth: int = 2):
        import concurrent.futures as cf
        import os
        if num_workers is None:
            num_workers = int(np.clip(int(3 / 4 * (os.cpu_count() or 16)), 1, 2 * batch_size))   # cell_detection.py:270-273
        self.wsi, self.ids, self.bs, self.dev = wsi, tile_ids, batch_size, device
        self.pool = cf.ThreadPoolExecutor(max_workers=num_workers)
        self.depth = depth
        self.batches = [tile_ids[i:i + batch_size] for i in range(0, len(tile_ids), batch_size)]
        self._pending: List = []
        self._next = 0

    def _submit(self):
        ids = self.batches[self._next]
        self._next += 1
        names = [self.wsi.patches_list[i] for i in ids]
        futs = [self.pool.submit(self.wsi.load_patch_image, n) for n in names]
        mds = [self.pool.submit(self.wsi.load_patch_metadata, n) for n in names]
        self._pending.append((ids, futs, mds))

    def __iter__(self):
        while self._next < len(self.batches) and len(self._pending) < self.depth:
            self._submit()
        while self._pending:
            ids, futs, mds = self._pending.pop(0)
            imgs = [f.result() for f in futs]
            host = torch.empty((len(imgs),) + imgs[0].shape, dtype=torch.uint8, pin_memory=self.dev.type == "cuda")
            for i, im in enumerate(imgs):
                np.copyto(host[i].numpy(), im)
            if self._next < len(self.batches):
                self._submit()
            yield ids, host.to(self.dev, non_blocking=True), [m.result() for m in mds]
        self.pool.shutdown(wait=False)


def check_wsi(wsi: PatchedSlide, magnification: float = 40.0) -> None:
    """cell_detection.py:1009-1039."""
    assert wsi.metadata["magnification"] == magnification, "The slide must be patched at the network magnification"
    assert wsi.metadata["patch_size"] == 1024, "The patch-size must be 1024 (for 40x)"
    assert wsi.metadata["patch_overlap"] == 64, "The patch-overlap must be 64 pixels"


class SlideCells:
    """Columnar store of the cells of a slide (or of one rank's shard): the packed record arrays of
    cellvit_amd.sharding (tile coordinates) + one pooled token row per cell.  Cells become Python dicts only at the
    writer (`to_dicts`), never inside the tile loop."""

    def __init__(self, ir=None, fr=None, ct=None, tokens: Optional[torch.Tensor] = None):
        self.ir = np.zeros((0, S.N_ICOL), np.int32) if ir is None else ir
        self.fr = np.zeros((0, S.N_FCOL), np.float64) if fr is None else fr
        self.ct = np.zeros((0, 2), np.int32) if ct is None else ct
        self.tokens = tokens

    def __len__(self):
        return len(self.ir)

    @staticmethod
    def from_tile_records(rec: np.ndarray, pts: np.ndarray, tile: int, row: int, col: int, background: int,
                          patch_size: int = 1024, overlap: int = 64):
        """One tile's device records (structured array, _lib.REC_DTYPE) -> (ir, fr, ct, kept record slots).
        Drops what the reference drops: instances without a contour (post_proc:113-116) and background-type cells
        (cell_detection.py:354-355)."""
        keep = np.nonzero((rec["contour_len"] >= 3) & (rec["type"] != background))[0]
        r = rec[keep]
        bbox = np.stack([r["rmin"], r["cmin"], r["rmax"], r["cmax"]], 1).astype(np.int32).reshape(-1, 4)
        ir = np.zeros((len(r), S.N_ICOL), np.int32)
        ir[:, S.I_ROW], ir[:, S.I_COL], ir[:, S.I_TILE] = row, col, tile
        ir[:, S.I_RMIN:S.I_CMAX + 1] = bbox
        ir[:, S.I_TYPE], ir[:, S.I_ID], ir[:, S.I_CLEN] = r["type"], r["id"], r["contour_len"]
        ir[:, S.I_STATUS] = S.cell_status_array(bbox, patch_size, overlap)
        ir[:, S.I_EDGE] = S.cell_edge_array(bbox, patch_size)
        fr = np.stack([r["cx"], r["cy"], r["type_prob"]], 1).astype(np.float64).reshape(-1, S.N_FCOL)
        if len(r):
            idx = np.concatenate([np.arange(o, o + n) for o, n in zip(r["contour_off"], r["contour_len"])])
            ct = pts[idx].astype(np.int32)
        else:
            ct = np.zeros((0, 2), np.int32)
        return ir, fr, ct, keep

    @staticmethod
    def concat(parts: List["SlideCells"]) -> "SlideCells":
        if not parts:
            return SlideCells()
        toks = [p.tokens for p in parts if p.tokens is not None]
        return SlideCells(np.concatenate([p.ir for p in parts]), np.concatenate([p.fr for p in parts]),
                          np.concatenate([p.ct for p in parts]), torch.cat(toks) if toks else None)

    def contour_slices(self):
        lens = self.ir[:, S.I_CLEN].astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]) if len(lens) else np.zeros(0, np.int64)
        return offs, lens

    def select(self, idx: np.ndarray) -> "SlideCells":
        offs, lens = self.contour_slices()
        parts = [self.ct[offs[i]:offs[i] + lens[i]] for i in idx]
        ct = np.concatenate(parts).astype(np.int32) if parts else np.zeros((0, 2), np.int32)
        tok = self.tokens[torch.as_tensor(idx, dtype=torch.long, device=self.tokens.device)] if self.tokens is not None else None
        return SlideCells(self.ir[idx], self.fr[idx], ct, tok)

    def to_dicts(self, patch_size: int, downsampling: float, overlap: int) -> List[dict]:
        """Cell dicts of `cells.json` in global slide coordinates (cell_detection.py:341-391)."""
        offs, lens = self.contour_slices()
        out = []
        for k in range(len(self.ir)):
            i, f = self.ir[k], self.fr[k]
            row, col = int(i[S.I_ROW]), int(i[S.I_COL])
            xg, yg = S.global_offset(row, col, patch_size, downsampling, overlap)
            off = np.array([xg, yg])
            bbox = np.array([[i[S.I_RMIN], i[S.I_CMIN]], [i[S.I_RMAX], i[S.I_CMAX]]])
            d = {
                "bbox": (bbox + off).tolist(),
                "centroid": (f[[S.F_CX, S.F_CY]] + np.flip(off)).tolist(),
                "contour": (self.ct[offs[k]:offs[k] + lens[k]] + np.flip(off)).tolist(),
                "type_prob": float(f[S.F_PROB]), "type": int(i[S.I_TYPE]),
                "patch_coordinates": [row, col],
                "cell_status": int(i[S.I_STATUS]),
                "offset_global": off.tolist(),
            }
            if i[S.I_EDGE]:
                pos = S.cell_edge_position(bbox, patch_size)
                d["edge_position"] = True
                d["edge_information"] = {"position": pos, "edge_patches": S.edge_patches(pos, row, col)}
            else:
                d["edge_position"] = False
            out.append(d)
        return out


def finalize_slide(local: SlideCells, patch_size: int, downsampling: float, overlap: int, device=None,
                   logger: Optional[logging.Logger] = None) -> Tuple[SlideCells, List[dict]]:
    """Slide-level step after the tile loop (cell_detection.py:423-433), identical for any world size:
      1. every rank contributes ONLY its margin-cell records (status != 0) to one all-gatherv
         (`sharding.all_gather_margin_records`: RCCL over xGMI with device buffers, gloo on CPU);
      2. the gathered records are put in slide order (tile index) and ONE global `stitch_cells` runs — the same
         deterministic computation on every rank, so no second collective is needed;
      3. each rank keeps its mid cells + its surviving margin cells; the survivors of all ranks are then gathered
         in slide order for the single writer (rank 0).
    Returns (all kept cells of the slide in slide order, their dicts) — complete on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    dev = device or torch.device("cpu")
    is_margin = local.ir[:, S.I_STATUS] != 0
    m_idx = np.nonzero(is_margin)[0]
    margin = local.select(m_idx)
    margin.tokens = None
    # uid of a cell = (tile, id): instance ids are unique per tile
    gi, gf, gc = S.all_gather_margin_records(margin.ir, margin.fr, margin.ct, device=dev)
    perm = S.canonical_order(gi)
    gi, gf, gc = S.reorder_records(gi, gf, gc, perm)
    g = SlideCells(gi, gf, gc)
    g_dicts = g.to_dicts(patch_size, downsampling, overlap)
    keep_g = stitch_cells(g_dicts, logger)                       # indices into the global margin list (all status != 0)
    kept_uid = {(int(gi[k, S.I_TILE]), int(gi[k, S.I_ID])) for k in keep_g}
    keep_local = np.array([k for k in range(len(local))
                           if not is_margin[k] or (int(local.ir[k, S.I_TILE]), int(local.ir[k, S.I_ID])) in kept_uid],
                          dtype=np.int64)
    mine = local.select(keep_local)
    if world > 1:
        ai, af, ac = S.all_gather_margin_records(mine.ir, mine.fr, mine.ct, device=dev)   # the writer's gather (same packed format)
        tok = S.all_gather_rows(mine.tokens.to(dev)) if mine.tokens is not None else None
        perm = S.canonical_order(ai)
        ai, af, ac = S.reorder_records(ai, af, ac, perm)
        if tok is not None:
            tok = tok[torch.as_tensor(perm, dtype=torch.long, device=tok.device)]
        allc = SlideCells(ai, af, ac, tok)
    else:
        perm = S.canonical_order(mine.ir)
        allc = mine.select(perm)
    if logger:
        logger.info(f"[rank {rank}] cells after cleaning: {len(allc)} (margin cells exchanged: {len(gi)})")
    return allc, allc.to_dicts(patch_size, downsampling, overlap)


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
        self.mean = tuple(float(v) for v in norm.get("mean", (0.5, 0.5, 0.5)))
        self.std = tuple(float(v) for v in norm.get("std", (0.5, 0.5, 0.5)))
        self.pool_cap = 2048       # fixed token-pooling slots per tile; tiles with more records take the exact-size pass

    def _normalize(self, tiles_u8: torch.Tensor) -> torch.Tensor:
        """T.ToTensor + T.Normalize (:214-227) as a stand-alone device op: [B,H,W,3] u8 -> [B,3,H,W] f32.  The tile
        loop does not call this (cv_forward_u8 evaluates the same arithmetic inside the forward's loaders)."""
        import ctypes as C
        from .. import _lib
        t = tiles_u8.to(self.device).contiguous()
        B, H, W, _ = t.shape
        out = torch.empty((B, 3, H, W), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().cv_op_normalize_u8(t.data_ptr(), (C.c_float * 3)(*self.mean), (C.c_float * 3)(*self.std),
                                                      out.data_ptr(), B, H, W,
                                                      C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    # ------------------------------------------------------------------------------------------
    def run_tiles(self, wsi: PatchedSlide, tile_ids: List[int], batch_size: int, patch_size: int = 1024,
                  overlap: int = 64, num_workers: Optional[int] = None) -> Tuple[SlideCells, List[str], dict]:
        """The tile loop (cell_detection.py:306-421) for one rank's tiles.  Per batch: raw u8 tiles -> forward (HIP) ->
        post-processing on the argmax planes the forward wrote (HIP) -> cell-token pooling (HIP); then ONE device->host
        copy of the record / contour arrays.  Host work of batch k (array unpacking) overlaps the GPU work of k+1."""
        nuclei_types = self.run_conf["dataset_config"]["nuclei_types"]
        obj, ks = _params(int(wsi.metadata["magnification"]))
        parts: List[SlideCells] = []
        processed: List[str] = []
        stats = {"tiles": 0, "t_loop": 0.0}
        import time

        def enqueue(ids, x_u8, mds):
            pred = self.model.forward_u8(x_u8, self.mean, self.std, retrieve_tokens=True)
            bin_am, typ_am = self.model._last_argmax              # argmax planes written by the forward kernels
            inst, recs, n_recs, contours, n_pts = postprocess_device(bin_am, typ_am, pred["hv_map"],
                                                                     self.model.num_nuclei_classes, obj, ks)
            pooled, cap = pool_cell_tokens_fixed(pred["tokens"], recs, n_recs, self.model.patch_size, cap=self.pool_cap)
            ev = torch.cuda.Event()
            ev.record()
            return ids, mds, recs, n_recs, contours, n_pts, pooled, cap, ev, pred["tokens"]

        def finish(job):
            ids, mds, recs, n_recs, contours, n_pts, pooled, cap, ev, tokens = job
            ev.synchronize()
            nr, npt = n_recs.cpu().numpy(), n_pts.cpu().numpy()
            check_capacity(recs, nr, contours, npt)
            if (nr > cap).any():                                   # rare: a tile with more cells than fixed pooling slots
                exact, off = pool_cell_tokens(tokens, recs, n_recs, self.model.patch_size)
                pooled = [exact[int(off[b]):int(off[b]) + int(nr[b])] for b in range(len(ids))]   # per-tile rows, as pooled[b] below
            mx_r, mx_p = int(nr.max()), int(npt.max())
            rec_h = recs[:, :mx_r].cpu().numpy().view(REC_DTYPE).reshape(len(ids), mx_r)
            pts_h = contours[:, :mx_p].cpu().numpy()
            for b, (tile, md) in enumerate(zip(ids, mds)):
                row, col = int(md["row"]), int(md["col"])
                processed.append(f"{row}_{col}")
                ir, fr, ct, keep = SlideCells.from_tile_records(rec_h[b, :nr[b]], pts_h[b], tile, row, col,
                                                                nuclei_types["Background"], patch_size, overlap)
                sel = torch.as_tensor(keep, dtype=torch.long, device=pooled[b].device)
                parts.append(SlideCells(ir, fr, ct, pooled[b].index_select(0, sel)))
            stats["tiles"] += len(ids)

        t0 = time.perf_counter()
        pending = None
        with torch.no_grad(), torch.cuda.device(self.device):
            for ids, x_u8, mds in TilePrefetcher(wsi, tile_ids, batch_size, self.device, num_workers):
                job = enqueue(ids, x_u8, mds)
                if pending is not None:
                    finish(pending)
                pending = job
            if pending is not None:
                finish(pending)
        stats["t_loop"] = time.perf_counter() - t0
        return SlideCells.concat(parts), processed, stats

    def process_wsi(self, wsi: PatchedSlide, subdir_name: Optional[str] = None, patch_size: int = 1024,
                    overlap: int = 64, batch_size: int = 8, geojson: bool = False) -> dict:
        import torch.distributed as dist
        dd = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank() if dd else 0
        world = dist.get_world_size() if dd else 1
        nuclei_types = self.run_conf["dataset_config"]["nuclei_types"]
        outdir = Path(wsi.patched_slide_path) / "cell_detection" / (subdir_name or "")
        outdir.mkdir(exist_ok=True, parents=True)
        my_tiles = S.shard_tiles(len(wsi.patches_list), rank, world, block=batch_size)
        local, processed, stats = self.run_tiles(wsi, my_tiles, batch_size, patch_size, overlap)
        self.logger.info(f"[rank {rank}/{world}] {stats['tiles']} tiles in {stats['t_loop']:.2f} s "
                         f"({stats['tiles'] / max(stats['t_loop'], 1e-9):.1f} tiles/s), cells before cleaning: {len(local)}")
        exch_dev = self.device if (dd and dist.get_backend() == "nccl") else torch.device("cpu")
        allc, cells_all = finalize_slide(local, wsi.metadata["patch_size"], wsi.metadata["downsampling"], overlap,
                                         device=exch_dev, logger=self.logger)
        if world > 1:
            gathered: List[Optional[list]] = [None] * world
            dist.all_gather_object(gathered, processed)       # tile names only (a few bytes per tile)
            order = {f"{m['row']}_{m['col']}": i for i, m in enumerate(wsi.all_patch_metadata[n] for n in wsi.patches_list)}
            processed = sorted((p for part in gathered for p in part), key=lambda k: order.get(k, 1 << 30))
        if rank == 0:
            write_outputs(outdir, wsi.metadata, processed, nuclei_types, allc, cells_all, geojson)
        stats.update({"n_cells": len(cells_all), "outdir": str(outdir)})
        return stats


def pool_cell_tokens_fixed(tokens: torch.Tensor, recs: torch.Tensor, n_recs: torch.Tensor, patch_size: int,
                           cap: int = 2048):
    """cv_pool_tokens into a fixed [B, cap, D] buffer (row offset b*cap): needs no host-visible count, so the launch
    does not synchronise.  Returns (fp32 [B, cap, D], cap); tiles with more than `cap` records use pool_cell_tokens."""
    import ctypes as C
    from .. import _lib
    B, D, gh, gw = tokens.shape
    tok_nhwc = tokens.permute(0, 2, 3, 1).contiguous().float()
    cap = min(cap, recs.shape[1])
    out = torch.empty((B, cap, D), device=tokens.device, dtype=torch.float32)
    off = torch.arange(B, device=tokens.device, dtype=torch.int64) * cap
    with torch.cuda.device(tokens.device):
        _lib.check(_lib.load().cv_pool_tokens(tok_nhwc.data_ptr(), B, gh, gw, D, int(patch_size), recs.data_ptr(), recs.shape[1],
                                              n_recs.data_ptr(), off.data_ptr(), cap, out.data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream(tokens.device).cuda_stream)))
    return out, cap


def write_outputs(outdir: Path, wsi_metadata: dict, processed: List[str], nuclei_types: dict, allc: SlideCells,
                  cells_all: List[dict], geojson: bool) -> None:
    """The writers of cell_detection.py:438-475: cells.json, cell_detection.json, optional geojson pair, cells.pt."""
    from ..datamodel import make_cell_graph
    meta = {"wsi_metadata": wsi_metadata, "processed_patches": processed, "type_map": nuclei_types}
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
    if len(cells_all):
        D = allc.tokens.shape[1] if allc.tokens is not None else 0
        x = allc.tokens.float().cpu() if allc.tokens is not None else torch.zeros((len(cells_all), D))
        graph = make_cell_graph(
            x=x, positions=torch.stack([torch.Tensor(c["centroid"]) for c in cells_all]),
            contours=[torch.Tensor(c["contour"]) for c in cells_all],
            metadata={"wsi_metadata": wsi_metadata, "nuclei_types": nuclei_types})
        torch.save(graph, outdir / "cells.pt")


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
    import uuid
    feats = []
    for t in sorted(by_type):                       # `detected_types = sorted(df.type.unique())` (:560, 582)
        geoms = by_type[t]
        feats.append({
            "type": "Feature", "id": str(uuid.uuid4()),
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
