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
import ctypes as C
import json
import logging
import math
from collections import defaultdict
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import yaml

from .. import _lib
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
        ct = S.gather_segments(self.ct.reshape(-1, 2), offs, lens, idx).astype(np.int32) if len(idx) else np.zeros((0, 2), np.int32)
        tok = self.tokens[torch.as_tensor(idx, dtype=torch.long, device=self.tokens.device)] if self.tokens is not None else None
        return SlideCells(self.ir[idx], self.fr[idx], ct, tok)

    def geometry(self, patch_size: int, downsampling: float, overlap: int) -> dict:
        """The slide-coordinate arrays of every cell (cell_detection.py:341-391), computed on the whole arrays:
        bbox i64 [n,4] (rows += x_global, cols += y_global), centroid f64 [n,2] and contour i64 [m,2] ((x, y) += flipped
        offset), contour offsets i64 [n+1], offset_global i64 [n,2], edge u8 [n], edge position u8 [n,4] (get_cell_position)."""
        from .stitch import tile_offsets
        n = len(self.ir)
        ir, fr = self.ir, self.fr
        offs, lens = self.contour_slices()
        xg, yg = tile_offsets(ir[:, S.I_ROW], ir[:, S.I_COL], patch_size, downsampling, overlap)
        off = np.stack([xg, yg], 1).reshape(n, 2)                       # `offset_global` = [x_global, y_global] (:351)
        bbox_t = ir[:, S.I_RMIN:S.I_CMAX + 1].astype(np.int64)
        rep = np.repeat(np.arange(n), lens)
        ct_off = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=ct_off[1:])
        return {
            "bbox": np.ascontiguousarray(bbox_t + np.concatenate([off, off], 1)),
            "centroid": np.ascontiguousarray(fr[:, [S.F_CX, S.F_CY]] + off[:, ::-1]),   # (x, y): + flip(offset) (:352-353)
            "contour": np.ascontiguousarray(self.ct.astype(np.int64).reshape(-1, 2) + off[rep][:, ::-1]),
            "ct_off": ct_off, "offset_global": np.ascontiguousarray(off),
            "edge": np.ascontiguousarray((ir[:, S.I_EDGE] != 0).astype(np.uint8)),
            "edge_pos": np.ascontiguousarray(np.stack([bbox_t[:, 0] == 0, bbox_t[:, 3] == patch_size, bbox_t[:, 2] == patch_size,
                                                       bbox_t[:, 1] == 0], 1).astype(np.uint8)),       # get_cell_position (:787-817)
        }

    def to_dicts(self, patch_size: int, downsampling: float, overlap: int) -> List[dict]:
        """Cell dicts of `cells.json` in global slide coordinates (cell_detection.py:341-391).  The arithmetic runs on the
        whole arrays (`geometry`); the per-cell work is one dict literal over pre-converted Python lists."""
        n = len(self.ir)
        if n == 0:
            return []
        ir, fr = self.ir, self.fr
        g = self.geometry(patch_size, downsampling, overlap)
        bbox = g["bbox"].reshape(n, 2, 2).tolist()
        cent, ctg = g["centroid"].tolist(), g["contour"].tolist()
        prob, typ = fr[:, S.F_PROB].tolist(), ir[:, S.I_TYPE].tolist()
        rows, cols, status = ir[:, S.I_ROW].tolist(), ir[:, S.I_COL].tolist(), ir[:, S.I_STATUS].tolist()
        edge = g["edge"].astype(bool).tolist()
        off_l, offs_l = g["offset_global"].tolist(), g["ct_off"].tolist()
        pos = g["edge_pos"].astype(np.int64).tolist()
        out = []
        for k in range(n):
            d = {"bbox": bbox[k], "centroid": cent[k], "contour": ctg[offs_l[k]:offs_l[k + 1]], "type_prob": prob[k], "type": typ[k],
                 "patch_coordinates": [rows[k], cols[k]], "cell_status": status[k], "offset_global": off_l[k]}
            if edge[k]:
                d["edge_position"] = True
                d["edge_information"] = {"position": pos[k], "edge_patches": S.edge_patches(pos[k], rows[k], cols[k])}
            else:
                d["edge_position"] = False
            out.append(d)
        return out


def finalize_slide(local: SlideCells, patch_size: int, downsampling: float, overlap: int, device=None,
                   logger: Optional[logging.Logger] = None, compute_device=None,
                   timings: Optional[dict] = None, want_dicts: bool = True,
                   gather_to: Optional[int] = None) -> Tuple[Optional[SlideCells], Optional[List[dict]]]:
    """Slide-level step after the tile loop (cell_detection.py:423-433), identical for any world size:
      1. every rank contributes ONLY its margin-cell records (status != 0) to one all-gatherv
         (`sharding.all_gather_margin_records`: RCCL over xGMI with device buffers, gloo on CPU);
      2. the gathered records are put in slide order (tile index) and ONE global de-duplication runs on the packed arrays
         (`stitch.stitch_margin_records`: candidate pairs + exact polygon intersections on the GPU, the greedy rounds in the
         library's host code) — the same deterministic computation on every rank, so no second collective is needed;
      3. each rank keeps its mid cells + its surviving margin cells; the survivors of all ranks are then collected in slide
         order: `gather_to=r` (what the CLI does, r = 0 = the one writer) sends every rank's kept cells and token rows to rank r
         ONLY — point to point, exact sizes, nothing replicated: a rank's traffic is its own cells once (≈ 5 KB per cell with
         1280-float tokens; the writer receives the slide's ≈ 3 GB once instead of every rank receiving it);
         `gather_to=None` all-gathers them so that the result is complete on every rank (tests, callers that ask for it).
    Every collective is entered by every rank, whatever it holds (a rank may have received no tile at all).
    `device`: where the exchange buffers live (cuda under nccl, cpu under gloo); `compute_device`: where the geometry of the
    de-duplication runs (default: `device`).  `timings` (optional dict) receives the seconds of exchange / stitch / dicts.
    Returns (all kept cells of the slide in slide order, their dicts or None with want_dicts=False) — complete on every
    rank, or on rank `gather_to` only ((None, None) elsewhere).  Nothing on this path needs per-cell dicts: the writers render the files from the arrays (`write_outputs`)."""
    import time
    import torch.distributed as dist
    from .stitch import stitch_margin_records
    t_start = time.perf_counter()
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    dev = device or torch.device("cpu")
    is_margin = local.ir[:, S.I_STATUS] != 0
    m_idx = np.nonzero(is_margin)[0]
    margin = local.select(m_idx)
    margin.tokens = None
    gi, gf, gc = S.all_gather_margin_records(margin.ir, margin.fr, margin.ct, device=dev)
    perm = S.canonical_order(gi)
    gi, gf, gc = S.reorder_records(gi, gf, gc, perm)
    t_gather = time.perf_counter()
    keep_g = stitch_margin_records(gi, gc, patch_size, downsampling, overlap, device=compute_device or dev, logger=logger)
    t_stitch = time.perf_counter()
    # uid of a cell = (tile, id): instance ids are unique per tile
    uid = lambda ir: ir[:, S.I_TILE].astype(np.int64) * (1 << 32) + ir[:, S.I_ID].astype(np.int64)   # noqa: E731
    keep_local = np.nonzero(~is_margin | np.isin(uid(local.ir), uid(gi[keep_g])))[0].astype(np.int64)
    mine = local.select(keep_local)
    n_total = sum(S.all_gather_int(len(mine), dev)) if world > 1 else len(mine)
    sent = recv = 0
    if world > 1 and gather_to is not None:
        # the writer's gather: kept cells + their token rows travel to ONE rank (only one writer exists, cell_detection.py:423-475);
        # every rank sends exactly its own bytes — nothing is replicated
        got, s_, r_ = S.gather_records_to(mine.ir, mine.fr, mine.ct, gather_to, device=dev)
        sent += s_; recv += r_
        D = max(S.all_gather_int(int(mine.tokens.shape[1]) if mine.tokens is not None else 0, dev))
        tok = None
        if D > 0:
            t_loc = mine.tokens if mine.tokens is not None else torch.zeros((0, D), dtype=torch.float32)
            tok, s_, r_ = S.gather_rows_to(t_loc.to(dev).float(), gather_to)
            sent += s_; recv += r_
        if got is not None:
            ai, af, ac = got
            perm = S.canonical_order(ai)
            ai, af, ac = S.reorder_records(ai, af, ac, perm)
            if tok is not None:
                tok = tok[torch.as_tensor(perm, dtype=torch.long, device=tok.device)]
            allc = SlideCells(ai, af, ac, tok)
        else:
            allc = None
    elif world > 1:
        ai, af, ac = S.all_gather_margin_records(mine.ir, mine.fr, mine.ct, device=dev)   # complete on every rank (tests, callers that ask)
        # token rows: the width is agreed first, ranks without cells contribute [0, D]
        D = max(S.all_gather_int(int(mine.tokens.shape[1]) if mine.tokens is not None else 0, dev))
        tok = None
        if D > 0:
            t_loc = mine.tokens if mine.tokens is not None else torch.zeros((0, D), dtype=torch.float32)
            tok = S.all_gather_rows(t_loc.to(dev).float())
        perm = S.canonical_order(ai)
        ai, af, ac = S.reorder_records(ai, af, ac, perm)
        if tok is not None:
            tok = tok[torch.as_tensor(perm, dtype=torch.long, device=tok.device)]
        allc = SlideCells(ai, af, ac, tok)
    else:
        perm = S.canonical_order(mine.ir)
        allc = mine.select(perm)
    if logger:
        logger.info(f"[rank {rank}] cells after cleaning: {n_total} (margin cells exchanged: {len(gi)})")
    t_collect = time.perf_counter()
    dicts = allc.to_dicts(patch_size, downsampling, overlap) if (want_dicts and allc is not None) else None
    if timings is not None:
        margin_bytes = int(gi.nbytes + gf.nbytes + gc.nbytes)
        timings.update({"margin_records": int(len(gi)), "margin_kept": int(len(keep_g)), "n_cells_total": int(n_total),
                        "margin_bytes_all_gathered": margin_bytes, "writer_gather_bytes_sent": int(sent), "writer_gather_bytes_received": int(recv),
                        "exchange_s": (t_gather - t_start) + (t_collect - t_stitch), "stitch_s": t_stitch - t_gather,
                        "to_dicts_s": time.perf_counter() - t_collect})
    return allc, dicts


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
        # The post-processing of batch k runs on a second stream, released when the forward of batch k + 1 reaches its full-resolution decoder
        # stages (cv_stream_wait_stage): beside the encoder's persistent whole-CU GEMM workgroups the latency-bound chain only time-slices, beside
        # the decoder's short two-per-CU workgroups it interleaves (profiles/r06_f_pp_stage_ab.txt).  False: forward and post-processing of a
        # batch back to back on one stream (the reference's order, cell_detection.py:306-421).  Same results either way.
        self.overlap_postproc = True

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
                  overlap: int = 64, num_workers: Optional[int] = None, tail=None) -> Tuple[SlideCells, List[str], dict]:
        """The tile loop (cell_detection.py:306-421) for one rank's tiles.  Per batch: raw u8 tiles -> forward (HIP) ->
        post-processing on the argmax planes the forward wrote (HIP) -> cell-token pooling (HIP); then the record / contour
        arrays go to pinned host buffers on a copy stream.  Host work of batch k (array unpacking) overlaps the GPU work of k+1.
        With `overlap_postproc` (default) the post-processing + pooling of batch k run on a second stream, released when the forward of
        batch k+1 reaches its full-resolution decoder stages (cv_stream_wait_stage); the last batch's chain follows its own forward.
        `tail` (a `tail.SlideTail`): every finished batch is handed to the streaming slide tail — token rows leave the device per
        batch, geometry and JSON text are prepared by its worker threads while the loop runs; the returned cells then carry no tokens."""
        nuclei_types = self.run_conf["dataset_config"]["nuclei_types"]
        obj, ks = _params(int(wsi.metadata["magnification"]))
        parts: List[SlideCells] = []
        processed: List[str] = []
        stats = {"tiles": 0, "t_loop": 0.0}
        import time

        # Device -> host hand-over of a batch's record / contour arrays: asynchronous copies into pinned buffers on a SECOND stream
        # that waits only for that batch's own event.  (Copies issued on the compute stream are ordered behind the NEXT batch's
        # kernels, which were enqueued first: the host then waited a whole batch for them and unpacked with the GPU idle —
        # measured: 86 tiles/s through this loop against 97 for the same kernels in bench.py.)
        copy_stream = torch.cuda.Stream(self.device)
        pinned: List[Optional[tuple]] = [None, None]          # two sets: batch k is unpacked while batch k+1 is copied
        turn = [0]

        def pinned_like(i, *tensors):
            have = pinned[i]
            if have is None or any(h.shape != t.shape or h.dtype != t.dtype for h, t in zip(have, tensors)):
                have = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors)
                pinned[i] = have
            return have

        pp_stream = torch.cuda.Stream(self.device) if getattr(self, "overlap_postproc", True) else None
        armed = [False]

        def forward_only(ids, x_u8, mds):
            pred = self.model.forward_u8(x_u8, self.mean, self.std, retrieve_tokens=True)
            if pp_stream is not None and not armed[0]:            # from the next forward on, the engine records its stage events
                e = getattr(self.model, "_last_engine", None)
                if e is not None:
                    _lib.check(e.lib.cv_stream_wait_stage(e.h, 0, None))
                    armed[0] = True
            return ids, mds, pred, self.model._last_argmax        # argmax planes written by the forward kernels

        def enqueue(fw, staged):
            """Post-processing, token pooling and the device -> host copies of one forwarded batch.  staged: the NEXT batch's forward has been
            enqueued behind it — the chain is released when that forward reaches its full-resolution stages; else behind everything enqueued."""
            ids, mds, pred, (bin_am, typ_am) = fw
            compute = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(pp_stream if pp_stream is not None else compute):
                if pp_stream is not None:
                    if staged and armed[0]:
                        e = self.model._last_engine
                        _lib.check(e.lib.cv_stream_wait_stage(e.h, 2, C.c_void_p(pp_stream.cuda_stream)))
                    else:                                          # (last batch, or a model without stage events)
                        pp_stream.wait_stream(compute)
                    for t_ in (bin_am, typ_am, pred["hv_map"], pred["tokens"]):     # allocated on the compute stream, read on this one
                        t_.record_stream(pp_stream)
                inst, recs, n_recs, contours, n_pts = postprocess_device(bin_am, typ_am, pred["hv_map"],
                                                                         self.model.num_nuclei_classes, obj, ks)
                pooled, cap = pool_cell_tokens_fixed(pred["tokens"], recs, n_recs, self.model.patch_size, cap=self.pool_cap)
                ev = torch.cuda.Event()
                ev.record()
                if pp_stream is not None:                          # finish() may run the exact-size pooling pass on the compute stream
                    for t_ in (recs, n_recs):
                        t_.record_stream(compute)
            host = pinned_like(turn[0], recs, n_recs, contours, n_pts)
            turn[0] ^= 1
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ev)
                for h, t in zip(host, (recs, n_recs, contours, n_pts)):
                    h.copy_(t, non_blocking=True)
                ev_copy = torch.cuda.Event()
                ev_copy.record(copy_stream)
            return ids, mds, recs, n_recs, contours, n_pts, pooled, cap, ev_copy, pred["tokens"], host

        def finish(job):
            ids, mds, recs, n_recs, contours, n_pts, pooled, cap, ev_copy, tokens, host = job
            ev_copy.synchronize()                                  # this batch's copies only — the next batch keeps the GPU busy
            recs_h, nr_h, pts_all, npt_h = host
            nr, npt = nr_h.numpy().copy(), npt_h.numpy().copy()
            check_capacity(recs, nr, contours, npt)
            if (nr > cap).any():                                   # rare: a tile with more cells than fixed pooling slots
                exact, off = pool_cell_tokens(tokens, recs, n_recs, self.model.patch_size)
                pooled = [exact[int(off[b]):int(off[b]) + int(nr[b])] for b in range(len(ids))]   # per-tile rows
                copy_stream.wait_stream(torch.cuda.current_stream(self.device))     # (this pass ran on the compute stream; recs / tokens are complete:
                                                                                    #  this batch's copies, behind its whole chain, were waited for above)
            mx_r, mx_p = int(nr.max()), int(npt.max())
            rec_h = recs_h[:, :mx_r].numpy().view(REC_DTYPE).reshape(len(ids), mx_r)
            pts_h = pts_all[:, :mx_p].numpy()
            tiles_out, rows = [], []
            for b, (tile, md) in enumerate(zip(ids, mds)):
                row, col = int(md["row"]), int(md["col"])
                processed.append(f"{row}_{col}")
                ir, fr, ct, keep = SlideCells.from_tile_records(rec_h[b, :nr[b]], pts_h[b], tile, row, col,
                                                                nuclei_types["Background"], patch_size, overlap)
                tiles_out.append((ir, fr, ct, len(keep)))
                rows.append(keep if isinstance(pooled, list) else keep + b * cap)
            # token rows of the kept cells: ONE gather per batch, on the copy stream.  (An index upload on the compute stream is
            # a synchronous copy ordered behind the next batch's kernels: the host sat out a whole batch there, then unpacked
            # with the GPU idle — 12 % of the loop.)
            with torch.cuda.stream(copy_stream):
                if isinstance(pooled, list):
                    tok = torch.cat([pooled[b].index_select(0, torch.as_tensor(rows[b], dtype=torch.long, device=pooled[b].device))
                                     for b in range(len(ids))]) if len(ids) else None
                else:
                    idx = torch.as_tensor(np.concatenate(rows) if rows else np.zeros(0, np.int64), dtype=torch.long, device=pooled.device)
                    tok = pooled.reshape(-1, pooled.shape[-1]).index_select(0, idx)
                # cross-stream lifetimes for the caching allocator: `pooled` was allocated on the post-processing (or compute) stream and is
                # read here on the copy stream (the job is dropped right after); `tok` is allocated on the copy stream and read later by
                # torch.cat / select on the compute stream
                for t_ in (pooled if isinstance(pooled, list) else [pooled]):
                    t_.record_stream(copy_stream)
                if tok is not None:
                    tok.record_stream(torch.cuda.current_stream(self.device))
            if tail is not None:
                if tiles_out:
                    tail.add_batch(ids[0], np.concatenate([t[0] for t in tiles_out]), np.concatenate([t[1] for t in tiles_out]),
                                   np.concatenate([t[2] for t in tiles_out]).reshape(-1, 2), tok, copy_stream)
                for ir, fr, ct, n in tiles_out:
                    parts.append(SlideCells(ir, fr, ct, None))
            else:
                o = 0
                for ir, fr, ct, n in tiles_out:
                    parts.append(SlideCells(ir, fr, ct, tok[o:o + n]))
                    o += n
            stats["tiles"] += len(ids)

        t0 = time.perf_counter()
        pending = None
        with torch.no_grad(), torch.cuda.device(self.device):
            prev_fw = None
            for ids, x_u8, mds in TilePrefetcher(wsi, tile_ids, batch_size, self.device, num_workers):
                fw = forward_only(ids, x_u8, mds)
                if pp_stream is None:
                    job = enqueue(fw, False)
                elif prev_fw is not None:
                    job = enqueue(prev_fw, True)                   # the previous batch's chain, beside this batch's decoder
                else:
                    job = None
                prev_fw = fw if pp_stream is not None else None
                if job is not None:
                    if pending is not None:
                        finish(pending)
                    pending = job
            if prev_fw is not None:                                # the last batch: nothing follows it
                job = enqueue(prev_fw, False)
                if pending is not None:
                    finish(pending)
                pending = job
            if pending is not None:
                finish(pending)
            copy_stream.synchronize()
            if pp_stream is not None:
                pp_stream.synchronize()
        stats["t_loop"] = time.perf_counter() - t0
        if tail is not None:      # the tail holds the batches; the caller only counts the cells
            ir = np.concatenate([p.ir for p in parts]) if parts else None
            return SlideCells(ir), processed, stats
        return SlideCells.concat(parts), processed, stats

    def wait_for_writers(self) -> None:
        """Join the writer thread of the previous slide (see `process_wsi(..., defer_write=True)`); re-raises its error."""
        th = getattr(self, "_writer", None)
        if th is not None:
            th.join()
            self._writer = None
            err = getattr(self, "_writer_error", None)
            self._writer_error = None
            if err is not None:
                raise err

    def _agree_on_writer_error(self, exch_dev) -> None:
        import torch.distributed as dist
        err = None
        try:
            self.wait_for_writers()
        except BaseException as e:      # noqa: BLE001
            err = e
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=exch_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) and err is None:
                raise RuntimeError("the writer of the previous slide failed on rank 0")
        if err is not None:
            raise err

    def process_wsi(self, wsi: PatchedSlide, subdir_name: Optional[str] = None, patch_size: int = 1024,
                    overlap: int = 64, batch_size: int = 8, geojson: bool = False, defer_write: bool = False,
                    stream_tail: bool = True) -> dict:
        """`defer_write`: hand the files of this slide to a writer thread and return (process_dataset: the files of slide k are
        written — mostly outside the GIL — while the tile loop of slide k+1 runs); `wait_for_writers()` joins it.
        `stream_tail` (default): the streaming slide tail (`tail.SlideTail`) — token rows, geometry and JSON text of every batch are
        prepared while the tile loop runs, after it only margin records are exchanged and the files are assembled from chunks;
        False: the batch route (`finalize_slide` + `write_outputs`), which is also what the tests check the streaming route against."""
        import time
        import torch.distributed as dist
        dd = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank() if dd else 0
        world = dist.get_world_size() if dd else 1
        nuclei_types = self.run_conf["dataset_config"]["nuclei_types"]
        outdir = Path(wsi.patched_slide_path) / "cell_detection" / (subdir_name or "")
        outdir.mkdir(exist_ok=True, parents=True)
        my_tiles = S.shard_tiles(len(wsi.patches_list), rank, world, block=batch_size)
        exch_dev = self.device if (dd and dist.get_backend() == "nccl") else torch.device("cpu")
        mps, mds = wsi.metadata["patch_size"], wsi.metadata["downsampling"]
        tail = None
        t_enter = time.perf_counter()
        if stream_tail:
            from .tail import SlideTail
            # (remote ranks' geometry is re-gathered as records for the optional geojson pair: only a single-rank run renders it from the tail)
            tail = SlideTail(mps, mds, overlap, self.device, keep_geometry=geojson and world == 1)
        try:      # from here to the writer hand-off the tail owns native text buffers and worker threads: release them on ANY exit
            local, processed, stats = self.run_tiles(wsi, my_tiles, batch_size, patch_size, overlap, tail=tail)
            t_tiles = time.perf_counter()
            self.logger.info(f"[rank {rank}/{world}] {stats['tiles']} tiles in {stats['t_loop']:.2f} s "
                             f"({stats['tiles'] / max(stats['t_loop'], 1e-9):.1f} tiles/s), cells before cleaning: {len(local)}")
            timings: dict = {}
            # a writer failure of the PREVIOUS slide (rank 0 only) is agreed on by all ranks before this slide's first collective:
            # everyone aborts together instead of the other ranks hanging in the next exchange
            self._agree_on_writer_error(exch_dev)
            if tail is None:
                allc, _ = finalize_slide(local, mps, mds, overlap, device=exch_dev, logger=self.logger, compute_device=self.device,
                                         timings=timings, want_dicts=False, gather_to=0)
                job = None
            else:
                job = self._finish_streamed(tail, exch_dev, mps, mds, overlap, geojson, timings)
        except BaseException:
            if tail is not None:
                tail.close()
            raise
        if world > 1:
            gathered: List[Optional[list]] = [None] * world
            dist.all_gather_object(gathered, processed)       # tile names only (a few bytes per tile)
            order = {f"{m['row']}_{m['col']}": i for i, m in enumerate(wsi.all_patch_metadata[n] for n in wsi.patches_list)}
            processed = sorted((p for part in gathered for p in part), key=lambda k: order.get(k, 1 << 30))
        t0 = time.perf_counter()
        self.wait_for_writers()                                   # at most one slide's files in flight
        if rank == 0:
            if tail is None:
                wargs = (outdir, wsi.metadata, processed, nuclei_types, allc, geojson, mps, mds, overlap)
                target = lambda: write_outputs(*wargs)     # noqa: E731
            else:
                target = lambda: write_outputs_streamed(outdir, wsi.metadata, processed, nuclei_types, job, geojson, tail)   # noqa: E731
            if defer_write:
                import threading

                def _write():
                    try:
                        target()
                    except BaseException as e:      # noqa: BLE001  (re-raised by wait_for_writers)
                        self._writer_error = e
                self._writer = threading.Thread(target=_write, name="cellvit-writers")
                self._writer.start()
            else:
                target()
        elif tail is not None:
            tail.close()
        timings["write_s"] = time.perf_counter() - t0
        # where the wall time outside the tile loop went: before / after the loop inside run_tiles, agreement + exchange + stitch + gather, writers
        timings["tail_breakdown_s"] = {"run_tiles_outside_loop": (t_tiles - t_enter) - stats["t_loop"], "finish": t0 - t_tiles, "write": timings["write_s"]}
        stats.update({"n_cells": int(timings.get("n_cells_total", len(allc) if (tail is None and allc is not None) else 0)),
                      "cells_before_cleaning": len(local), "outdir": str(outdir), **timings})
        return stats

    def _finish_streamed(self, tail, exch_dev, patch_size, downsampling, overlap, geojson, timings: dict) -> Optional[dict]:
        return finish_streamed(tail, exch_dev, self.device, patch_size, downsampling, overlap, geojson, timings, self.logger)


def finish_streamed(tail, exch_dev, compute_device, patch_size, downsampling, overlap, geojson, timings: dict, logger=None) -> Optional[dict]:
    """After the tile loop, streaming route: exchange ONLY the margin records (one all-gatherv), one global de-duplication (the same
    deterministic computation on every rank), keep masks per batch, then the writer's gather of the kept chunks (point to point, exact
    sizes; nothing on one rank).  Returns the writer's job (chunk lists in slide order) on rank 0, None elsewhere."""
    import time
    import torch.distributed as dist
    from .stitch import stitch_margin_records
    dd = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if dd else 1
    rank = dist.get_rank() if dd else 0
    t_start = time.perf_counter()
    mi, mf, mc = tail.local_margin()                                   # (waits for the batches' worker jobs)
    t_ready = time.perf_counter()
    gi, gf, gc = S.all_gather_margin_records(mi, mf, mc, device=exch_dev)
    perm = S.canonical_order(gi)
    gi, gf, gc = S.reorder_records(gi, gf, gc, perm)
    t_gather = time.perf_counter()
    keep_g = stitch_margin_records(gi, gc, patch_size, downsampling, overlap, device=compute_device or exch_dev, logger=logger)
    t_stitch = time.perf_counter()
    n_mine = tail.set_survivors(gi[keep_g])
    n_total = sum(S.all_gather_int(n_mine, exch_dev)) if world > 1 else n_mine
    D = max(S.all_gather_int(tail.token_dim, exch_dev)) if world > 1 else tail.token_dim
    sh = tail.kept_shard()
    sent = recv = 0
    shards = [sh]
    if world > 1:
        shards, sent, recv = _gather_shards_to(sh, D, 0, exch_dev)
    job = None
    if rank == 0:
        job = _merge_shards(shards, D)
        job["n"] = int(n_total)
    if geojson and world > 1:
        mine = SlideCells.concat([SlideCells(b.ir, b.fr, b.ct).select(np.nonzero(b.keep)[0]) for b in tail.batches]) if tail.batches else SlideCells()
        got, s_, r_ = S.gather_records_to(mine.ir, mine.fr, mine.ct, 0, device=exch_dev)
        sent += s_; recv += r_
        if got is not None:
            ai, af, ac = got
            p2 = S.canonical_order(ai)
            ai, af, ac = S.reorder_records(ai, af, ac, p2)
            job["geo_cells"] = SlideCells(ai, af, ac, None)
    t_collect = time.perf_counter()
    if logger:
        logger.info(f"[rank {rank}] cells after cleaning: {n_total} (margin cells exchanged: {len(gi)})")
    timings.update({"margin_records": int(len(gi)), "margin_kept": int(len(keep_g)), "n_cells_total": int(n_total),
                    "margin_bytes_all_gathered": int(gi.nbytes + gf.nbytes + gc.nbytes), "writer_gather_bytes_sent": int(sent),
                    "writer_gather_bytes_received": int(recv), "tail_wait_workers_s": t_ready - t_start,
                    "exchange_s": (t_gather - t_ready) + (t_collect - t_stitch), "stitch_s": t_stitch - t_gather, "to_dicts_s": 0.0})
    return job


def _gather_shards_to(sh: dict, D: int, dst: int, dev):
    """The writer's gather of the streaming route: every rank's kept chunks, concatenated per rank (text bytes, token rows, positions,
    contour points, contour lengths) + per-batch keys / counts / text lengths, point to point to `dst`.  Returns (list of shards in
    rank order on `dst` — the writer's own shard stays as chunk lists with masks, the others are slices of the received tensors —,
    bytes sent, bytes received)."""
    import torch.distributed as dist
    rank = dist.get_rank()
    sent = recv = 0

    def g(t):
        nonlocal sent, recv
        parts, s_, r_ = S._gather_var_to(t.to(dev), dst)
        sent += s_; recv += r_
        return parts

    def cat_kept(pairs, width, dtype):
        rows = [(torch.from_numpy(a) if isinstance(a, np.ndarray) else a)[torch.from_numpy(k.astype(bool))] for a, k in pairs if a is not None and len(a)]
        return torch.cat(rows) if rows else torch.zeros((0, width), dtype=dtype)

    nb = len(sh["keys"])
    meta = torch.from_numpy(np.stack([sh["keys"], sh["counts"], np.asarray([len(c) for c in sh["text_cells"]], np.int64),
                                      np.asarray([len(c) for c in sh["text_det"]], np.int64),
                                      np.asarray([int(l.sum()) for l in sh["lens"]], np.int64)], 1).reshape(nb, 5))
    is_dst = rank == dst
    p_meta = g(meta)
    own = lambda t: torch.zeros((0,) + tuple(t.shape[1:]), dtype=t.dtype) if is_dst else t   # noqa: E731  (the writer keeps its own chunks in place)
    text0 = torch.from_numpy(np.concatenate(sh["text_cells"]) if nb else np.zeros(0, np.uint8))
    text1 = torch.from_numpy(np.concatenate(sh["text_det"]) if nb else np.zeros(0, np.uint8))
    p_t0, p_t1 = g(own(text0)), g(own(text1))
    p_tok = g(own(cat_kept(sh["tok"], D, torch.float32))) if D > 0 else None
    p_pos = g(own(cat_kept(sh["pos"], 2, torch.float32)))
    p_cont = g(own(cat_kept(sh["cont"], 2, torch.float32)))
    p_lens = g(own(torch.from_numpy(np.concatenate(sh["lens"]) if nb else np.zeros(0, np.int64))))
    if not is_dst:
        return None, sent, recv
    shards = []
    for r in range(dist.get_world_size()):
        if r == dst:
            shards.append(sh)
            continue
        m = p_meta[r].cpu().numpy().reshape(-1, 5)
        cnt, l0, l1, lc = m[:, 1], m[:, 2], m[:, 3], m[:, 4]
        t0, t1 = p_t0[r].cpu().numpy(), p_t1[r].cpu().numpy()
        tok = p_tok[r].cpu() if p_tok is not None else None
        pos, cont, lens = p_pos[r].cpu().numpy(), p_cont[r].cpu().numpy(), p_lens[r].cpu().numpy()
        o0, o1, oc, ol = (np.concatenate([[0], np.cumsum(a)]) for a in (l0, l1, cnt, lc))
        shards.append({"keys": m[:, 0], "counts": cnt,
                       "text_cells": [t0[o0[i]:o0[i + 1]] for i in range(len(m))], "text_det": [t1[o1[i]:o1[i + 1]] for i in range(len(m))],
                       "tok": [((tok[oc[i]:oc[i + 1]] if tok is not None else None), None) for i in range(len(m))],
                       "pos": [(pos[oc[i]:oc[i + 1]], None) for i in range(len(m))],
                       "cont": [(cont[ol[i]:ol[i + 1]], None) for i in range(len(m))],
                       "lens": [lens[oc[i]:oc[i + 1]] for i in range(len(m))]})
    return shards, sent, recv


def _merge_shards(shards: List[dict], D: int) -> dict:
    """All ranks' batches in slide order (first tile index of the batch): chunk lists for the writers."""
    items = []
    for sh in shards:
        for i in range(len(sh["keys"])):
            items.append((int(sh["keys"][i]), sh, i))
    items.sort(key=lambda t: t[0])
    pick = lambda name: [sh[name][i] for _, sh, i in items]   # noqa: E731
    lens = [sh["lens"][i] for _, sh, i in items]
    return {"D": D, "text_cells": pick("text_cells"), "text_det": pick("text_det"), "tok": pick("tok"), "pos": pick("pos"), "cont": pick("cont"),
            "lens": np.concatenate(lens).astype(np.int64) if lens else np.zeros(0, np.int64)}


def write_outputs_streamed(outdir: Path, wsi_metadata: dict, processed: List[str], nuclei_types: dict, job: dict, geojson: bool, tail) -> None:
    """The writers of cell_detection.py:438-475 from the streaming route's chunk lists: the two JSON documents are header + chunks + footer
    (byte for byte `cv_write_cells_json` of the kept cells), cells.pt a skip_data archive filled by parallel row writes (`tail.write_cells_pt_streamed`)."""
    import threading
    from . import tail as T
    try:
        meta = {"wsi_metadata": wsi_metadata, "processed_patches": processed, "type_map": nuclei_types}
        header = json.dumps(meta, default=_np_default)[1:-1].encode("utf-8")
        errors: List[BaseException] = []

        def guarded(fn, *a):
            try:
                fn(*a)
            except BaseException as e:      # noqa: BLE001
                errors.append(e)
        th = [threading.Thread(target=guarded, args=(T.write_json_chunks, outdir / "cells.json", header, job["text_cells"])),
              threading.Thread(target=guarded, args=(T.write_json_chunks, outdir / "cell_detection.json", header, job["text_det"]))]
        for t in th:
            t.start()
        n = int(job["n"])
        if n:
            T.write_cells_pt_streamed(outdir / "cells.pt", n, int(job["D"]), job["tok"] if job["D"] > 0 else [], job["pos"], job["cont"], job["lens"],
                                      {"wsi_metadata": wsi_metadata, "nuclei_types": nuclei_types})
        if geojson:
            if job.get("geo_cells") is not None:
                allc = job["geo_cells"]
                write_geojson_pair(outdir, allc.geometry(wsi_metadata["patch_size"], wsi_metadata["downsampling"], tail.ov),
                                   np.ascontiguousarray(allc.ir[:, S.I_TYPE].astype(np.int32)))
            else:
                cen, offs, cont, typ = [], [0], [], []
                for b in sorted(tail.batches, key=lambda b: b.key):
                    k = b.keep.astype(bool)
                    c0, o0, ct0, t0 = b.geo
                    lens = np.diff(o0)
                    cen.append(c0[k]); typ.append(t0[k]); cont.append(ct0[np.repeat(k, lens)])
                    offs.extend((offs[-1] + np.cumsum(lens[k])).tolist())
                g = {"centroid": np.ascontiguousarray(np.concatenate(cen)) if cen else np.zeros((0, 2)),
                     "ct_off": np.asarray(offs, np.int64), "contour": np.ascontiguousarray(np.concatenate(cont)) if cont else np.zeros((0, 2), np.int64)}
                write_geojson_pair(outdir, g, np.ascontiguousarray(np.concatenate(typ).astype(np.int32)) if typ else np.zeros(0, np.int32))
        for t in th:
            t.join()
        if errors:
            raise errors[0]
    finally:
        # several GB of host arrays and text: released by a helper thread (munmap of that much takes a few tenths of a second)
        job.clear()
        threading.Thread(target=tail.close, name="cellvit-tail-release").start()


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
                  geojson: bool, patch_size: int, downsampling: float, overlap: int) -> None:
    """The writers of cell_detection.py:438-475: cells.json, cell_detection.json, optional geojson pair, cells.pt.
    The two JSON files are rendered from the arrays by the library's host code (`cv_write_cells_json`, called through ctypes:
    the GIL is released) while this thread pickles cells.pt; per-cell dicts exist only for the optional geojson pair."""
    import ctypes as C
    import threading
    from .. import _lib
    from ..datamodel import save_cell_graph
    lib = _lib.load()
    n = len(allc)
    g = allc.geometry(patch_size, downsampling, overlap)
    meta = {"wsi_metadata": wsi_metadata, "processed_patches": processed, "type_map": nuclei_types}
    header = json.dumps(meta, default=_np_default)[1:-1].encode("utf-8")           # members without the outer braces
    typ = np.ascontiguousarray(allc.ir[:, S.I_TYPE].astype(np.int32))
    prob = np.ascontiguousarray(allc.fr[:, S.F_PROB].astype(np.float64))
    rc = np.ascontiguousarray(allc.ir[:, [S.I_ROW, S.I_COL]].astype(np.int32))
    status = np.ascontiguousarray(allc.ir[:, S.I_STATUS].astype(np.int32))
    errors: List[str] = []

    def write_json():
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        for name, det in (("cells.json", 0), ("cell_detection.json", 1)):
            rcode = lib.cv_write_cells_json(str(outdir / name).encode(), header, det, n, p(g["bbox"]), p(g["centroid"]), p(g["ct_off"]),
                                            p(g["contour"]), p(prob), p(typ), p(rc), p(status), p(g["offset_global"]), p(g["edge"]),
                                            p(g["edge_pos"]))
            if rcode != _lib.CV_OK:
                errors.append((lib.cv_last_error() or b"cv_write_cells_json failed").decode("utf-8", "replace"))
    th = threading.Thread(target=write_json)
    th.start()
    if n:
        D = allc.tokens.shape[1] if allc.tokens is not None else 0
        x = allc.tokens.float().cpu() if allc.tokens is not None else torch.zeros((n, D))
        # one [len_k, 2] float32 view per cell of ONE contour tensor (`torch.Tensor(list)` per cell in the reference); the list's
        # pickle records are generated in bulk (datamodel.save_cell_graph: 0.24 s instead of 2.8 s per 10^5 cells)
        save_cell_graph(outdir / "cells.pt", x, torch.from_numpy(g["centroid"].astype(np.float32)),
                        torch.from_numpy(g["contour"].astype(np.float32)), np.diff(g["ct_off"]).tolist(),
                        {"wsi_metadata": wsi_metadata, "nuclei_types": nuclei_types})
    if geojson:
        write_geojson_pair(outdir, g, typ)
    th.join()
    if errors:
        raise RuntimeError(errors[0])


def _np_default(o):
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, np.ndarray):
        return o.tolist()
    raise TypeError(type(o))


def write_geojson_pair(outdir: Path, g: dict, typ: np.ndarray) -> None:
    """cells.geojson (MultiPolygon per type) and cell_detection.geojson (MultiPoint per type) rendered from the arrays by the library's host code
    (`cv_write_geojson`); the features' frames — type, uuid, classification name / colour — are rendered here exactly as `convert_geojson` builds them."""
    import ctypes as C
    import uuid
    from .. import _lib
    lib = _lib.load()
    n = len(typ)
    present = sorted(int(t) for t in np.unique(typ)) if n else []
    ft = np.ascontiguousarray(np.asarray(present, np.int32))
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    for name, polygons in (("cells.geojson", 1), ("cell_detection.geojson", 0)):
        heads, tails = [], []
        for t in present:
            frame = {"type": "Feature", "id": str(uuid.uuid4()), "geometry": {"type": "MultiPolygon" if polygons else "MultiPoint", "coordinates": []},
                     "properties": {"objectType": "annotation",
                                    "classification": {"name": TYPE_NUCLEI_DICT.get(t, str(t)), "color": COLOR_DICT.get(t, [0, 0, 0])}}}
            text = json.dumps(frame)
            cut = text.index('"coordinates": [') + len('"coordinates": [')
            heads.append(text[:cut].encode()); tails.append(text[cut:].encode())
        H = (C.c_char_p * max(len(present), 1))(*heads)
        T = (C.c_char_p * max(len(present), 1))(*tails)
        _lib.check(lib.cv_write_geojson(str(outdir / name).encode(), polygons, n, p(g["centroid"]), p(g["ct_off"]), p(g["contour"]), p(typ),
                                        len(present), p(ft), H, T))


def convert_geojson(cell_list: List[dict], polygons: bool = False) -> List[dict]:
    """cell_detection.py:538-597 + template_geojson.py:9-52: one MultiPolygon / MultiPoint feature per cell type."""
    by_type: Dict[int, list] = defaultdict(list)
    for c in cell_list:
        if polygons:
            ring = [list(p) for p in c["contour"]]            # the integer contour lists as they are (:564-568: `c.append(c[0])`)
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
            inf.process_wsi(slide, conf["outdir_subdir"], batch_size=conf["batch_size"], geojson=conf["geojson"], defer_write=True)
        inf.wait_for_writers()
    else:
        raise ValueError("Unknown command")


if __name__ == "__main__":
    main()
