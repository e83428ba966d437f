"""Slide-level de-duplication of margin cells on packed record arrays — no per-cell dicts.

Reference: ``CellPostProcessor`` (/root/reference/cell_segmentation/inference/cell_detection.py:600-767): mid cells are
kept; edge cells (bbox touches the tile border) are dropped unless the neighbouring tile produced no margin cells
(:645-674); the remaining margin cells go through up to 20 rounds in which, of every group of cells whose contour
polygons overlap by more than 1 % of either area, the largest *other* cell survives (:676-767, shapely STRtree +
``Polygon.intersection``).

Here the geometry runs on the MI355X (``cv_stitch_overlaps``: bbox-grid candidate pairs, exact polygon areas and exact
polygon-intersection areas, one thread per candidate pair) and the rounds — a sequential sweep in cell order by
definition — in the library's host code (``cv_stitch_select``).  ``device=cpu`` (the gloo tests; no GPU) evaluates the same
pair list with the numpy polygon routines below: that route is the checker of the device kernels
(tests/test_gpu_stitch.py), not a production path.
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Optional, Tuple

import numpy as np
import torch

from .. import _lib
from .. import sharding as S

OVERLAP_FRACTION = 0.01        # cell_detection.py:728-735
MAX_ROUNDS = 20                # :690

# get_edge_patch (cell_detection.py:877-902): [top, right, down, left] -> FIRST neighbour tile offset (drow, dcol); patterns the
# reference's if-chain does not name yield None (the cell is then kept)
_FIRST_EDGE = {(1, 0, 0, 0): (-1, 0), (1, 1, 0, 0): (-1, 0), (0, 1, 0, 0): (0, 1), (0, 1, 1, 0): (0, 1),
               (0, 0, 1, 0): (1, 0), (0, 0, 1, 1): (1, 0), (0, 0, 0, 1): (0, -1), (1, 0, 0, 1): (0, -1)}
_EDGE_DR = np.zeros(16, np.int64)
_EDGE_DC = np.zeros(16, np.int64)
_EDGE_OK = np.zeros(16, bool)
for _pos, (_dr, _dc) in _FIRST_EDGE.items():
    _code = _pos[0] * 8 + _pos[1] * 4 + _pos[2] * 2 + _pos[3]
    _EDGE_DR[_code], _EDGE_DC[_code], _EDGE_OK[_code] = _dr, _dc, True


def tile_offsets(row: np.ndarray, col: np.ndarray, patch_size: int, downsampling: float, overlap: int):
    """Vectorised `sharding.global_offset` (cell_detection.py:341-350; `int()` truncates towards zero)."""
    row = row.astype(np.float64)
    col = col.astype(np.float64)
    xg = np.trunc(row * patch_size * downsampling - (row + 0.5) * overlap).astype(np.int64)
    yg = np.trunc(col * patch_size * downsampling - (col + 0.5) * overlap).astype(np.int64)
    return xg, yg


def global_geometry(ir: np.ndarray, ct: np.ndarray, patch_size: int, downsampling: float, overlap: int, offsets=None):
    """Slide-coordinate boxes and contours of packed records, exactly as `SlideCells.to_dicts` shifts them
    (cell_detection.py:351-359: rows += x_global, cols += y_global; contour (x, y) += (y_global, x_global)).
    `offsets` = (x_global, y_global) per record where the caller's tiling is not the WSI one (MoNuSeg: i * 256 - i * overlap).
    Returns (bbox int32 [n,4] = rmin, cmin, rmax, cmax; contour offsets int64 [n+1]; contour points int32 [m,2])."""
    n = len(ir)
    xg, yg = offsets if offsets is not None else tile_offsets(ir[:, S.I_ROW], ir[:, S.I_COL], patch_size, downsampling, overlap)
    xg, yg = np.asarray(xg, np.int64), np.asarray(yg, np.int64)
    bbox = ir[:, S.I_RMIN:S.I_CMAX + 1].astype(np.int64) + np.stack([xg, yg, xg, yg], 1)
    lens = ir[:, S.I_CLEN].astype(np.int64)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    rep = np.repeat(np.arange(n), lens)
    ctg = ct.astype(np.int64).reshape(-1, 2) + np.stack([yg[rep], xg[rep]], 1)
    return bbox.astype(np.int32), off, ctg.astype(np.int32)


def edge_rule(ir: np.ndarray, patch_size: int) -> np.ndarray:
    """`_clean_edge_cells` (cell_detection.py:645-674) on margin records: bool [n], True for the cells that enter the overlap
    removal — margin cells that do not touch the tile border, and border cells whose neighbouring tile (the FIRST entry of
    `edge_patches`) has no margin cell at all."""
    bb = ir[:, S.I_RMIN:S.I_CMAX + 1]
    edge = ir[:, S.I_EDGE] != 0
    code = ((bb[:, 0] == 0) * 8 + (bb[:, 3] == patch_size) * 4 + (bb[:, 2] == patch_size) * 2 + (bb[:, 1] == 0)).astype(np.int64)
    row, col = ir[:, S.I_ROW].astype(np.int64), ir[:, S.I_COL].astype(np.int64)
    existing = np.unique(row * (1 << 32) + col)
    nb = (row + _EDGE_DR[code]) * (1 << 32) + (col + _EDGE_DC[code])
    has_nb = _EDGE_OK[code] & np.isin(nb, existing)
    return ~edge | ~has_nb


# ---- polygon geometry on the host (checker of the device kernels; evaluates pairs the device hands back) ----------------------
def poly_area(contour: np.ndarray) -> float:
    """Area of the closed polygon through the contour points (shoelace), as `shapely.Polygon(contour).area`."""
    pts = np.asarray(contour, dtype=np.float64)
    if len(pts) < 3:
        return 0.0
    x, y = pts[:, 0], pts[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def _edge_crossings_y(a: np.ndarray, b: np.ndarray) -> np.ndarray:
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
    p0, p1 = poly, np.roll(poly, -1, axis=0)
    y0, y1 = p0[:, 1], p1[:, 1]
    hit = ((y0 <= yc) & (yc < y1)) | ((y1 <= yc) & (yc < y0))
    xs = p0[hit, 0] + (yc - y0[hit]) * (p1[hit, 0] - p0[hit, 0]) / (y1[hit] - y0[hit])
    return np.sort(xs)


def intersection_area(a: np.ndarray, b: np.ndarray) -> float:
    """EXACT area of the intersection of two polygons (even-odd interiors) by slab decomposition: between two consecutive
    event ordinates (vertices of either polygon, crossings of an a-edge with a b-edge) every interval end point is linear in
    y, so the common length L(y) is linear and the midpoint rule integrates it exactly."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
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


def candidate_pairs_host(bbox: np.ndarray) -> np.ndarray:
    """Pairs (i < j) of strictly overlapping boxes, int32 [p, 2] sorted by (i, j): uniform grid keyed on the top-left corner
    with a cell no smaller than the largest box, so that overlapping boxes lie in neighbouring grid cells."""
    n = len(bbox)
    if n < 2:
        return np.zeros((0, 2), np.int32)
    b = bbox.astype(np.int64)
    r0, c0, r1, c1 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    G = max(64, int(max((r1 - r0).max(), (c1 - c0).max())) + 1)
    gy, gx = (r0 - r0.min()) // G + 1, (c0 - c0.min()) // G + 1
    NX = int(gx.max()) + 2
    key = gy * NX + gx
    order = np.argsort(key, kind="stable")
    ks = key[order]
    out = []
    idx = np.arange(n)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            nk = key + dy * NX + dx
            lo, hi = np.searchsorted(ks, nk, "left"), np.searchsorted(ks, nk, "right")
            cnt = hi - lo
            tot = int(cnt.sum())
            if tot == 0:
                continue
            ii = np.repeat(idx, cnt)
            base = np.repeat(np.cumsum(cnt) - cnt, cnt)
            jj = order[np.repeat(lo, cnt) + (np.arange(tot) - base)]
            m = (jj > ii) & ~((r0[jj] >= r1[ii]) | (r1[jj] <= r0[ii]) | (c0[jj] >= c1[ii]) | (c1[jj] <= c0[ii]))
            out.append(np.stack([ii[m], jj[m]], 1))
    if not out:
        return np.zeros((0, 2), np.int32)
    p = np.concatenate(out)
    p = p[np.lexsort((p[:, 1], p[:, 0]))]
    return p.astype(np.int32)


def repair_rings(off: np.ndarray, ctg: np.ndarray, flags: Optional[np.ndarray] = None):
    """`if not poly.is_valid: poly.buffer(0)` -> the part of the largest area (cell_detection.py:689-704) on the packed contour
    arrays: rings whose lattice chain visits a lattice point twice (a blob pinched at a diagonal, a one-pixel spur) are replaced
    by their largest simple lobe — host code of the library (`cv_stitch_repair_rings`).  `flags` (u8 [n], from the device
    kernel `cv_stitch_ring_flags`) limits the work to the rings it names; None = every ring is examined.
    Returns (off, ctg, number of repaired rings); the inputs themselves when nothing was repaired."""
    n = len(off) - 1
    if n <= 0 or (flags is not None and not flags.any()):
        return off, ctg, 0
    lib = _lib.load()
    off_c = np.ascontiguousarray(off, np.int64)
    ct_c = np.ascontiguousarray(ctg, np.int32).reshape(-1, 2)
    out_off = np.empty(n + 1, np.int64)
    out_ct = np.empty((max(1, len(ct_c)), 2), np.int32)
    fl = None if flags is None else np.ascontiguousarray(flags, np.uint8)
    nrep = C.c_int32(0)
    _lib.check(lib.cv_stitch_repair_rings(off_c.ctypes.data, ct_c.ctypes.data, n, fl.ctypes.data if fl is not None else None,
                                          out_off.ctypes.data, out_ct.ctypes.data, C.byref(nrep)))
    if nrep.value == 0:
        return off, ctg, 0
    return out_off, out_ct[:int(out_off[-1])], int(nrep.value)


def overlaps_host(bbox: np.ndarray, off: np.ndarray, ctg: np.ndarray):
    """(pairs, inter, area) as `cv_stitch_overlaps`, on the host with the numpy routines above (rings repaired first)."""
    off, ctg, _ = repair_rings(off, ctg)
    pairs = candidate_pairs_host(bbox)
    n = len(bbox)
    area = np.array([poly_area(ctg[off[i]:off[i + 1]]) for i in range(n)], np.float64)
    inter = np.array([intersection_area(ctg[off[i]:off[i + 1]], ctg[off[j]:off[j + 1]]) if area[i] > 0 and area[j] > 0 else 0.0
                      for i, j in pairs], np.float64)
    return pairs, inter, area


def overlaps_device(bbox: np.ndarray, off: np.ndarray, ctg: np.ndarray, device: torch.device, cap: Optional[int] = None):
    """(pairs, inter, area) from the HIP kernels; pairs sorted by (i, j).  Pairs whose slabs exceeded the kernel's fixed
    capacities (inter == -1: outlines with > 24 simultaneous edges on one scan line) are evaluated by `intersection_area`."""
    n = len(bbox)
    if n == 0:
        return np.zeros((0, 2), np.int32), np.zeros(0), np.zeros(0)
    lib = _lib.load()
    cap = int(cap or max(1024, 16 * n))
    with torch.cuda.device(device):
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

        def upload(off_, ctg_):
            d_off_ = torch.from_numpy(np.ascontiguousarray(off_, np.int64)).to(device)
            d_ct_ = torch.from_numpy(np.ascontiguousarray(ctg_, np.int32).reshape(-1, 2)).to(device)
            if d_ct_.numel() == 0:
                d_ct_ = torch.zeros((1, 2), dtype=torch.int32, device=device)
            return d_off_, d_ct_

        d_bbox = torch.from_numpy(np.ascontiguousarray(bbox, np.int32)).to(device)
        d_off, d_ct = upload(off, ctg)
        # invalid rings (cell_detection.py:689-704): found on the device, repaired (a handful per slide) by the library's host code
        d_flags = torch.empty((n,), dtype=torch.uint8, device=device)
        _lib.check(lib.cv_stitch_ring_flags(d_off.data_ptr(), d_ct.data_ptr(), n, d_flags.data_ptr(), stream))
        flags = d_flags.cpu().numpy()
        if flags.any():
            off, ctg, nrep = repair_rings(off, ctg, flags)
            if nrep:
                d_off, d_ct = upload(off, ctg)
        d_area = torch.empty((n,), dtype=torch.float64, device=device)
        extent = (C.c_int32 * 4)(int(bbox[:, 0].min()), int(bbox[:, 1].min()), int(bbox[:, 2].max()), int(bbox[:, 3].max()))
        npairs = C.c_int32(0)
        for attempt in range(2):      # a capacity overflow returns the exact pair count: retry once with it (never lose the slide)
            d_pairs = torch.empty((cap, 2), dtype=torch.int32, device=device)
            d_inter = torch.empty((cap,), dtype=torch.float64, device=device)
            rc = lib.cv_stitch_overlaps(d_bbox.data_ptr(), d_off.data_ptr(), d_ct.data_ptr(), n, extent, d_pairs.data_ptr(),
                                        d_inter.data_ptr(), d_area.data_ptr(), cap, C.byref(npairs), stream)
            if rc == _lib.CV_ERR_SHAPE and attempt == 0 and int(npairs.value) > cap:
                cap = int(npairs.value)
                continue
            _lib.check(rc)
            break
        p = int(npairs.value)
        pairs = d_pairs[:p].cpu().numpy()
        inter = d_inter[:p].cpu().numpy()
        area = d_area.cpu().numpy()
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    pairs, inter = pairs[order], inter[order].copy()
    for k in np.nonzero(inter < 0)[0]:
        i, j = pairs[k]
        inter[k] = intersection_area(ctg[off[i]:off[i + 1]], ctg[off[j]:off[j + 1]])
    return pairs, inter, area


def select_rounds(pairs: np.ndarray, inter: np.ndarray, area: np.ndarray, alive: np.ndarray,
                  logger: Optional[logging.Logger] = None) -> np.ndarray:
    """The <= 20 greedy rounds (cell_detection.py:690-767) in the library's host code; returns the surviving mask."""
    lib = _lib.load()
    n = len(area)
    i, j = pairs[:, 0], pairs[:, 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        ov = ((area[i] > 0) & (area[j] > 0) & ((inter / area[i] > OVERLAP_FRACTION) | (inter / area[j] > OVERLAP_FRACTION)))
    pairs_c = np.ascontiguousarray(pairs, np.int32)
    ov_c = np.ascontiguousarray(ov, np.uint8)
    area_c = np.ascontiguousarray(area, np.float64)
    alive_c = np.ascontiguousarray(alive, np.uint8).copy()
    rounds = C.c_int32(0)
    counts = np.zeros(MAX_ROUNDS, np.int32)
    _lib.check(lib.cv_stitch_select(pairs_c.ctypes.data, ov_c.ctypes.data, len(pairs_c), area_c.ctypes.data, alive_c.ctypes.data,
                                    n, MAX_ROUNDS, C.byref(rounds), counts.ctypes.data))
    if logger:
        for r in range(int(rounds.value)):
            logger.info(f"Iteration {r}: Found overlap of # cells: {int(counts[r])}")
    return alive_c.astype(bool)


def stitch_margin_records(ir: np.ndarray, ct: np.ndarray, patch_size: int, downsampling: float, overlap: int,
                          device: Optional[torch.device] = None, logger: Optional[logging.Logger] = None, offsets=None) -> np.ndarray:
    """Indices (ascending) of the MARGIN records (`ir` holds only cells with status != 0, in slide order) that survive
    `CellPostProcessor.post_process_cells`."""
    n = len(ir)
    if n == 0:
        return np.zeros(0, np.int64)
    bbox, off, ctg = global_geometry(ir, ct, patch_size, downsampling, overlap, offsets)
    alive = edge_rule(ir, patch_size)
    if device is not None and device.type == "cuda":
        pairs, inter, area = overlaps_device(bbox, off, ctg, device)
    else:
        pairs, inter, area = overlaps_host(bbox, off, ctg)
    keep = select_rounds(pairs, inter, area, alive, logger)
    return np.nonzero(keep)[0].astype(np.int64)
