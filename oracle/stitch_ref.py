"""TEST INFRASTRUCTURE (oracle): slide-level de-duplication of cells on per-cell dicts — a direct restatement of
``CellPostProcessor`` (/root/reference/cell_segmentation/inference/cell_detection.py:600-767) with shapely-free polygon
geometry that is INDEPENDENT of the product's: intersection areas by Sutherland-Hodgman clipping against a triangle fan
(the product integrates over horizontal slabs), invalid rings repaired to their largest lobe on the lattice chain (the
reference: ``buffer(0)`` -> largest part).  Only tests/ import this module: it is the checker of the array / device
implementation in ``cellvit_amd/inference/stitch.py`` (the product path).  Parity with shapely / GEOS itself (part order of
``buffer(0)``, STRtree query order) is UNPINNED: shapely is not installable in this environment."""
from __future__ import annotations

import logging
from collections import defaultdict
from typing import Dict, List, Optional, Tuple

import numpy as np


def _poly_area(contour: np.ndarray) -> float:
    """Area of the closed polygon through the contour points (shoelace), as `shapely.Polygon(contour).area`."""
    pts = np.asarray(contour, dtype=np.float64)
    if len(pts) < 3:
        return 0.0
    x, y = pts[:, 0], pts[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


# ---- invalid rings: `Polygon.buffer(0)` -> largest part (cell_detection.py:689-704) -----------------------------------------
# The contours are outer borders traced on the pixel lattice (Suzuki-Abe, CHAIN_APPROX_SIMPLE): every edge runs along one of
# the 8 lattice directions and the ring never crosses itself, but it can TOUCH itself — a blob that is 8-connected through a
# diagonal pinch passes the pinch pixel twice, a one-pixel-wide spur is walked out and back.  shapely calls such a ring
# invalid ("Ring Self-intersection"); the reference repairs it with buffer(0), which returns the ring's simple lobes as a
# MultiPolygon (zero-width spurs vanish), and keeps the lobe of the largest area (np.argmax: the first of equal areas, in
# GEOS' part order — an order this restatement cannot know: equal lobes are resolved to the first one CLOSED along the ring).
# Restated without shapely: expand the ring to its lattice chain, cut a loop off whenever a lattice point is visited a
# second time, keep the loop with the largest |area|.  [parity with GEOS itself unpinned: shapely is not installable here]
def _lattice_chain(contour: np.ndarray) -> List[Tuple[int, int]]:
    pts = [(int(p[0]), int(p[1])) for p in np.asarray(contour)]
    out: List[Tuple[int, int]] = []
    n = len(pts)
    for k in range(n):
        (x0, y0), (x1, y1) = pts[k], pts[(k + 1) % n]
        dx, dy = x1 - x0, y1 - y0
        steps = max(abs(dx), abs(dy))
        if steps == 0:
            continue
        if not (dx == 0 or dy == 0 or abs(dx) == abs(dy)):       # not a lattice direction: keep the edge as it is
            out.append((x0, y0))
            continue
        sx, sy = (dx > 0) - (dx < 0), (dy > 0) - (dy < 0)
        for t in range(steps):
            out.append((x0 + t * sx, y0 + t * sy))
    return out


def _shoelace2(loop: List[Tuple[int, int]]) -> int:
    s = 0
    for k in range(len(loop)):
        (x0, y0), (x1, y1) = loop[k], loop[(k + 1) % len(loop)]
        s += x0 * y1 - x1 * y0
    return s


def ring_is_simple(contour: np.ndarray) -> bool:
    chain = _lattice_chain(contour)
    return len(set(chain)) == len(chain)


def largest_lobe(contour: np.ndarray) -> np.ndarray:
    """The ring itself when no lattice point repeats, else its largest simple lobe (see above), as an [m, 2] integer array of
    lattice points (collinear points are not removed: areas and intersections do not depend on them)."""
    chain = _lattice_chain(contour)
    if len(set(chain)) == len(chain):
        return np.asarray(contour)
    loops: List[List[Tuple[int, int]]] = []
    stack: List[Tuple[int, int]] = []
    pos: Dict[Tuple[int, int], int] = {}
    for pt in chain:
        if pt in pos:                                            # second visit: the path since the first visit is a closed loop
            k = pos[pt]
            loop = stack[k:]
            for q in loop[1:]:
                del pos[q]
            del stack[k + 1:]
            loops.append(loop)
        else:
            pos[pt] = len(stack)
            stack.append(pt)
    loops.append(stack)                                          # what remains closes through the ring's first point
    best, best_a = None, -1
    for lp in loops:
        a = abs(_shoelace2(lp)) if len(lp) >= 3 else 0
        if a > best_a:
            best, best_a = lp, a
    return np.asarray(best, dtype=np.int64).reshape(-1, 2)


# ---- intersection area: an INDEPENDENT method ------------------------------------------------------------------------------
# (the product — cellvit_amd/inference/stitch.py on the host, csrc/stitch.hip on the device — integrates the common length of
#  the two polygons over horizontal slabs.)  Here: polygon B is a signed sum of the triangles (b0, bj, bj+1) of a fan around its
# first vertex — 1_B = sum_j sign_j 1_Tj almost everywhere for a simple ring — and polygon A is clipped against each triangle
# with Sutherland-Hodgman (three half-plane passes; a concave subject yields degenerate bridges whose signed area cancels), so
#   area(A ∩ B) = | sum_j sign_j * shoelace(clip(A, Tj)) |.
def _clip_halfplane(poly: np.ndarray, p0: np.ndarray, p1: np.ndarray) -> np.ndarray:
    """Sutherland-Hodgman pass: the part of the (closed) vertex list `poly` on the left of / on the directed line p0 -> p1."""
    if len(poly) == 0:
        return poly
    d = p1 - p0
    side = d[0] * (poly[:, 1] - p0[1]) - d[1] * (poly[:, 0] - p0[0])        # >= 0: inside
    nxt = np.roll(poly, -1, axis=0)
    side_n = np.roll(side, -1)
    inside, inside_n = side >= 0, side_n >= 0
    cross = inside != inside_n
    with np.errstate(divide="ignore", invalid="ignore"):
        t = side / (side - side_n)
        inter = poly + t[:, None] * (nxt - poly)                             # (only the rows with cross == True are used)
    # per edge (v, v_next): emit v when it is inside, then the crossing point when the edge changes side
    cand = np.stack([poly, inter], axis=1)                                  # [n, 2, 2]
    keep = np.stack([inside, cross], axis=1)                                # [n, 2]
    return cand[keep]


def _signed_area(poly: np.ndarray) -> float:
    if len(poly) < 3:
        return 0.0
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def _intersection_area(a: np.ndarray, b: np.ndarray) -> float:
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if len(a) < 3 or len(b) < 3:
        return 0.0
    if max(a[:, 1].min(), b[:, 1].min()) >= min(a[:, 1].max(), b[:, 1].max()) or \
            max(a[:, 0].min(), b[:, 0].min()) >= min(a[:, 0].max(), b[:, 0].max()):
        return 0.0
    if _signed_area(a) < 0:
        a = a[::-1]
    total = 0.0
    for j in range(1, len(b) - 1):
        tri = np.stack([b[0], b[j], b[j + 1]])
        sgn = _signed_area(tri)
        if sgn == 0.0:
            continue
        if sgn < 0:
            tri = tri[::-1]
        piece = a
        for e in range(3):
            piece = _clip_halfplane(piece, tri[e], tri[(e + 1) % 3])
            if len(piece) < 3:
                break
        if len(piece) >= 3:
            total += (1.0 if sgn > 0 else -1.0) * _signed_area(piece)
    return abs(total)


def _overlap_fractions(ca: dict, cb: dict) -> Tuple[float, float, float, float]:
    """(intersection / area_a, intersection / area_b, area_a, area_b) of two cells' contour polygons — the quantities
    the reference takes from shapely (`cell_detection.py:722-747`), computed exactly (no shapely here)."""
    a, b = largest_lobe(np.asarray(ca["contour"])), largest_lobe(np.asarray(cb["contour"]))
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
            # np.argmax (cell_detection.py:743-746): the FIRST of equal areas, in the order the tree returned them — here ascending
            # cell index (the STRtree's own order is unpinned)
            out.append(i if not sub else sub[int(np.argmax([a for a, _ in sub]))][1])
            done.add(i)
        if logger:
            logger.info(f"Iteration {iteration}: Found overlap of # cells: {overlaps}")
        merged = sorted(set(out))
        if overlaps == 0:
            break
    return sorted(keep + merged)


