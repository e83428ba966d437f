"""Tile-parallel execution across the GPUs of one node (one process per GPU, torch.distributed over
RCCL/xGMI — backend "nccl" on ROCm — or gloo on CPU for tests).

The hot path has no cross-tile state (reference hot loop: cell_detection.py:306-421), so tiles shard
embarrassingly: a static block-cyclic split of the slide's row-major tile list, weights replicated,
NO collective on the data path.  The only exchange is at the slide level: cells whose bounding box
reaches into the 64-px overlap margin of their tile ("margin cells", get_cell_position_marging,
cell_detection.py:820-874) may be detected twice by neighbouring tiles that live on different ranks;
their fixed-size records are all-gathered (all_gather of counts, then of padded buffers — ~10-40 MB per
gigapixel slide, latency-bound) so that every rank (or rank 0) can run the de-duplication.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

# record layout (int32 columns / float64 columns); bbox / centroid / contour in TILE coordinates (the global offset is a
# function of (row, col), cell_detection.py:341-359, and is re-applied by whoever consumes the records)
I_ROW, I_COL, I_RMIN, I_CMIN, I_RMAX, I_CMAX, I_TYPE, I_STATUS, I_EDGE, I_TILE, I_CLEN, I_ID = range(12)
F_CX, F_CY, F_PROB = range(3)
N_ICOL, N_FCOL = 12, 3


def shard_tiles(n_tiles: int, rank: int, world: int, block: int = 1) -> List[int]:
    """Static block-cyclic assignment of tile indices (row-major order of wsi.patches_list) to ranks."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    out = []
    for start in range(rank * block, n_tiles, world * block):
        out.extend(range(start, min(start + block, n_tiles)))
    return out


def global_offset(row: int, col: int, patch_size: int, downsample: float, overlap: int) -> Tuple[int, int]:
    """cell_detection.py:341-350: (x_global, y_global) offset of a tile on the highest magnification.
    NB the reference names the ROW offset 'x_global' and adds the pair flipped to (x, y) quantities."""
    xg = int(row * patch_size * downsample - (row + 0.5) * overlap)
    yg = int(col * patch_size * downsample - (col + 0.5) * overlap)
    return xg, yg


def cell_edge_position(bbox: np.ndarray, patch_size: int = 1024) -> List[int]:
    """get_cell_position (cell_detection.py:787-817): [top, right, down, left] flags of a bbox
    [[rmin, cmin], [rmax, cmax]] touching the tile border."""
    return [int(bbox[0, 0] == 0), int(bbox[1, 1] == patch_size), int(bbox[1, 0] == patch_size), int(bbox[0, 1] == 0)]


def cell_status(bbox: np.ndarray, patch_size: int = 1024, margin: int = 64) -> int:
    """get_cell_position_marging (cell_detection.py:820-874): 0 = mid, 1..8 clockwise from top-left."""
    lo, hi = margin, patch_size - margin
    if not (np.max(bbox) > hi or np.min(bbox) < lo):
        return 0
    top, left = bbox[0, 0] < lo, bbox[0, 1] < lo
    down, right = bbox[1, 0] > hi, bbox[1, 1] > hi
    if top:
        return 1 if left else (3 if right else 2)
    if right:
        return 5 if down else 4
    if down:
        return 7 if left else 6
    if left:
        return 8
    return None   # unreachable for well-formed boxes (mirrors the reference's fall-through)


def cell_status_array(bbox: np.ndarray, patch_size: int = 1024, margin: int = 64) -> np.ndarray:
    """Vectorised `cell_status` for bbox [n, 4] = (rmin, cmin, rmax, cmax): int32 [n]."""
    bbox = np.asarray(bbox).reshape(-1, 4)
    lo, hi = margin, patch_size - margin
    mid = ~((bbox.max(1) > hi) | (bbox.min(1) < lo))
    top, left = bbox[:, 0] < lo, bbox[:, 1] < lo
    down, right = bbox[:, 2] > hi, bbox[:, 3] > hi
    st = np.select(
        [mid, top & left, top & right, top, right & down, right, down & left, down, left],
        [0, 1, 3, 2, 5, 4, 7, 6, 8], default=-1).astype(np.int32)
    return st


def cell_edge_array(bbox: np.ndarray, patch_size: int = 1024) -> np.ndarray:
    """`np.max(bbox) == patch_size or np.min(bbox) == 0` (cell_detection.py:380) for bbox [n, 4]: bool [n]."""
    bbox = np.asarray(bbox).reshape(-1, 4)
    return (bbox.max(1) == patch_size) | (bbox.min(1) == 0)


_EDGE_TABLE = {   # get_edge_patch (cell_detection.py:877-902): position -> neighbour tile offsets (drow, dcol)
    (1, 0, 0, 0): [(-1, 0)], (1, 1, 0, 0): [(-1, 0), (-1, 1), (0, 1)], (0, 1, 0, 0): [(0, 1)],
    (0, 1, 1, 0): [(0, 1), (1, 1), (1, 0)], (0, 0, 1, 0): [(1, 0)], (0, 0, 1, 1): [(1, 0), (1, -1), (0, -1)],
    (0, 0, 0, 1): [(0, -1)], (1, 0, 0, 1): [(0, -1), (-1, -1), (-1, 0)],
}


def edge_patches(position: Sequence[int], row: int, col: int):
    offs = _EDGE_TABLE.get(tuple(int(p) for p in position))
    return None if offs is None else [[row + dr, col + dc] for dr, dc in offs]


def pack_margin_records(tile_dict: dict, row: int, col: int, patch_size: int = 1024, margin: int = 64):
    """Per-tile nucleus dict (post_proc:126-151 layout) -> (int32 [n,12], float64 [n,3], int32 [m,2]) arrays of
    the cells that are NOT safely in the tile centre (status != 0) — the only ones that need an exchange."""
    irows, frows, contours = [], [], []
    for cid, c in tile_dict.items():
        st = cell_status(c["bbox"], patch_size, margin)
        if st == 0:
            continue
        bb = c["bbox"]
        edge = int(np.max(bb) == patch_size or np.min(bb) == 0)
        cont = c["contour"] if c.get("contour") is not None else np.zeros((0, 2), np.int32)
        irows.append([row, col, bb[0, 0], bb[0, 1], bb[1, 0], bb[1, 1], c["type"], st, edge, c.get("tile", 0), len(cont), cid])
        frows.append([c["centroid"][0], c["centroid"][1], c["type_prob"]])
        contours.append(np.asarray(cont, np.int32).reshape(-1, 2))
    ir = np.asarray(irows, np.int32).reshape(-1, N_ICOL)
    fr = np.asarray(frows, np.float64).reshape(-1, N_FCOL)
    ct = np.concatenate(contours).astype(np.int32) if contours else np.zeros((0, 2), np.int32)
    return ir, fr, ct


def _all_gather_var(t: torch.Tensor, group=None) -> List[torch.Tensor]:
    """all-gatherv: gather counts, pad to the maximum, gather, trim.  Works on nccl (device tensors) and gloo."""
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts + [1])
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:c] for b, c in zip(bufs, counts)]


def all_gather_margin_records(ir: np.ndarray, fr: np.ndarray, ct: np.ndarray, device=None, group=None):
    """Exchange the margin-cell records of all ranks (rank order preserved).  Returns the concatenated
    (int32 [N,12], float64 [N,3], int32 [M,2]) arrays; contour slices follow the record order.
    `device`: where the exchange buffers live — a cuda device under backend "nccl" (RCCL over xGMI), cpu under gloo."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return ir, fr, ct
    dev = device or torch.device("cpu")
    parts_i = _all_gather_var(torch.from_numpy(np.ascontiguousarray(ir)).to(dev), group)
    parts_f = _all_gather_var(torch.from_numpy(np.ascontiguousarray(fr)).to(dev), group)
    parts_c = _all_gather_var(torch.from_numpy(np.ascontiguousarray(ct)).to(dev), group)
    cat = lambda ps: torch.cat(ps).cpu().numpy()   # noqa: E731
    return cat(parts_i), cat(parts_f), cat(parts_c)


def _gather_var_to(t: torch.Tensor, dst: int, group=None):
    """gatherv to ONE rank: the row counts travel in an all-gather of one integer per rank, the rows point to point — every
    rank sends exactly its own bytes to `dst` and nothing is replicated (an all-gatherv pads every contribution to the largest
    and delivers world x that to every rank).  Returns (list of per-rank tensors on `dst`, None elsewhere; bytes sent by this
    rank; bytes received by this rank)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    t = t.contiguous()
    row_bytes = t.element_size() * int(np.prod(t.shape[1:])) if t.dim() > 1 else t.element_size()
    if rank != dst:
        if counts[rank] > 0:
            dist.send(t, dst, group=group)
        return None, counts[rank] * row_bytes, 0
    parts, recv = [], 0
    for r in range(world):
        if r == rank:
            parts.append(t)
            continue
        buf = torch.empty((counts[r],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if counts[r] > 0:
            dist.recv(buf, src=r, group=group)
            recv += counts[r] * row_bytes
        parts.append(buf)
    return parts, 0, recv


def gather_records_to(ir: np.ndarray, fr: np.ndarray, ct: np.ndarray, dst: int = 0, device=None, group=None):
    """The packed records of all ranks on rank `dst` only (rank order preserved) — the writer's gather: only one rank writes the
    slide's files (cell_detection.py:423-475).  Returns ((ir, fr, ct) on `dst`, None elsewhere; bytes sent; bytes received)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return (ir, fr, ct), 0, 0
    dev = device or torch.device("cpu")
    sent = recv = 0
    out = []
    for a in (ir, fr, ct):
        parts, s_, r_ = _gather_var_to(torch.from_numpy(np.ascontiguousarray(a)).to(dev), dst, group)
        sent += s_; recv += r_
        out.append(torch.cat(parts).cpu().numpy() if parts is not None else None)
    return (tuple(out) if out[0] is not None else None), sent, recv


def gather_rows_to(t: torch.Tensor, dst: int = 0, group=None):
    """gatherv of a [n, ...] tensor along dim 0 to rank `dst` (rank order).  Returns (tensor on `dst` / None, bytes sent, received)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t, 0, 0
    parts, s_, r_ = _gather_var_to(t.contiguous(), dst, group)
    return (torch.cat(parts) if parts is not None else None), s_, r_


def all_gather_int(v: int, device=None, group=None) -> List[int]:
    """One integer from every rank (rank order); [v] without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [int(v)]
    t = torch.tensor([int(v)], dtype=torch.int64, device=device or torch.device("cpu"))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [int(o.item()) for o in out]


def all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """all-gatherv of a [n, ...] tensor along dim 0 (rank order); identity without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    return torch.cat(_all_gather_var(t.contiguous(), group))


def canonical_order(ir: np.ndarray) -> np.ndarray:
    """Permutation that sorts gathered records by tile index (stable): the order a single process walking the slide's
    row-major tile list produces, whatever the world size and shard layout were."""
    return np.argsort(ir[:, I_TILE], kind="stable")


def gather_segments(flat: np.ndarray, offs: np.ndarray, lens: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """Rows [offs[i], offs[i] + lens[i]) of `flat` for i in idx, concatenated in that order — one index array instead of one slice
    per cell (10^5 - 10^6 cells per slide)."""
    idx = np.asarray(idx, np.int64)
    if idx.size == 0:
        return flat[:0].copy()
    l = np.asarray(lens, np.int64)[idx]
    total = int(l.sum())
    if total == 0:
        return flat[:0].copy()
    out_start = np.cumsum(l) - l
    pos = np.arange(total, dtype=np.int64) + np.repeat(np.asarray(offs, np.int64)[idx] - out_start, l)
    return flat[pos]


def reorder_records(ir: np.ndarray, fr: np.ndarray, ct: np.ndarray, perm: np.ndarray):
    """Apply a record permutation to (ir, fr) and rebuild the flat contour array in the new record order."""
    lens = ir[:, I_CLEN].astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]) if len(lens) else np.zeros(0, np.int64)
    ct2 = gather_segments(ct.reshape(-1, 2), offs, lens, perm).astype(np.int32) if len(perm) else np.zeros((0, 2), np.int32)
    return ir[perm], fr[perm], ct2
