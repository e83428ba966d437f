"""Seeded synthetic post-processing inputs (SURVEY §8d 'Synthetic inputs — post-proc').

Random-weight logits are salt-and-pepper and exercise none of the nucleus-separation logic, so the
post-processing parity tests and the benchmark use maps synthesised the way the training targets
are defined (HV map = per-instance horizontal/vertical offset from the centre of mass, each side
normalised to [-1, 1]; cf. /root/reference/cell_segmentation/datasets/pannuke.py:335-415)."""
from __future__ import annotations

import numpy as np


def synth_nuclei_maps(tile_idx: int, size: int = 1024, n_cells: int = 800, n_types: int = 6, noise: float = 0.03):
    """Returns (type_map u8 [H,W], binary_map u8 [H,W], hv float32 [2,H,W], gt instance map int32)."""
    rng = np.random.default_rng(7 + tile_idx)
    H = W = size
    inst = np.zeros((H, W), dtype=np.int32)
    tmap = np.zeros((H, W), dtype=np.uint8)
    n = max(1, int(n_cells * (size / 1024.0) ** 2))
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(1, n + 1):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        ra, rb = rng.uniform(7, 16, 2)
        th = rng.uniform(0, np.pi)
        R = int(max(ra, rb)) + 2
        y0, y1 = max(int(cy) - R, 0), min(int(cy) + R + 1, H)
        x0, x1 = max(int(cx) - R, 0), min(int(cx) + R + 1, W)
        if y0 >= y1 or x0 >= x1:
            continue
        dy = yy[y0:y1, x0:x1] - cy
        dx = xx[y0:y1, x0:x1] - cx
        u = dx * np.cos(th) + dy * np.sin(th)
        v = -dx * np.sin(th) + dy * np.cos(th)
        m = (u / ra) ** 2 + (v / rb) ** 2 <= 1.0
        # at most ~30 % overlap with what is already painted
        prev = inst[y0:y1, x0:x1][m]
        if prev.size == 0 or (prev > 0).mean() > 0.3:
            continue
        inst[y0:y1, x0:x1][m] = i
        tmap[y0:y1, x0:x1][m] = rng.integers(1, n_types)
    hv = np.zeros((2, H, W), dtype=np.float32)
    ids = np.unique(inst)
    ids = ids[ids > 0]
    for i in ids:
        ys, xs = np.nonzero(inst == i)
        cy, cx = int(ys.mean() + 0.5), int(xs.mean() + 0.5)
        ox = (xs - cx).astype(np.float32)
        oy = (ys - cy).astype(np.float32)
        for o in (ox, oy):
            neg, pos = o < 0, o > 0
            if neg.any():
                o[neg] /= -o[neg].min()
            if pos.any():
                o[pos] /= o[pos].max()
        hv[0, ys, xs] = ox
        hv[1, ys, xs] = oy
    fg = inst > 0
    hv += np.where(fg, noise, 0.02).astype(np.float32) * rng.standard_normal(hv.shape).astype(np.float32)
    flip = rng.random((H, W)) < 0.10
    tnoise = rng.integers(0, n_types, size=(H, W)).astype(np.uint8)
    tmap = np.where(flip & fg, tnoise, tmap).astype(np.uint8)
    return tmap, fg.astype(np.uint8), hv.astype(np.float32), inst


def synth_world_maps(seed: int, size: int = 1920, n_cells: int = 2800, n_types: int = 6, noise: float = 0.03,
                     return_inst: bool = False):
    """A PERIODIC nucleus world (torus of `size` pixels): (type_map u8, binary_map u8, hv float32 [2,H,W]) whose crops at
    any offset (numpy `take(..., mode='wrap')`) are mutually consistent — neighbouring slide tiles cut from it see the SAME
    nuclei in their 64-px overlap, which is what the slide-level de-duplication needs as input.  Same
    construction as `synth_nuclei_maps` (ellipses, HV = offset from the centre of mass normalised per side, noise), every
    step on the instance's own window."""
    rng = np.random.default_rng(seed)
    H = W = size
    inst = np.zeros((H, W), dtype=np.int32)
    tmap = np.zeros((H, W), dtype=np.uint8)
    hv = np.zeros((2, H, W), dtype=np.float32)
    wins = {}
    for i in range(1, n_cells + 1):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        ra, rb = rng.uniform(7, 16, 2)
        th = rng.uniform(0, np.pi)
        R = int(max(ra, rb)) + 2
        ys = np.arange(int(cy) - R, int(cy) + R + 1)
        xs = np.arange(int(cx) - R, int(cx) + R + 1)
        dy = (ys - cy)[:, None]
        dx = (xs - cx)[None, :]
        u = dx * np.cos(th) + dy * np.sin(th)
        v = -dx * np.sin(th) + dy * np.cos(th)
        m = (u / ra) ** 2 + (v / rb) ** 2 <= 1.0
        yw, xw = np.ix_(ys % H, xs % W)
        prev = inst[yw, xw][m]
        if prev.size == 0 or (prev > 0).mean() > 0.3:
            continue
        sub = inst[yw, xw]
        sub[m] = i
        inst[yw, xw] = sub
        subt = tmap[yw, xw]
        subt[m] = rng.integers(1, n_types)
        tmap[yw, xw] = subt
        wins[i] = (ys, xs)
    for i, (ys, xs) in wins.items():
        yw, xw = np.ix_(ys % H, xs % W)
        m = inst[yw, xw] == i
        if not m.any():
            continue
        py, px = np.nonzero(m)
        cy, cx = int(py.mean() + 0.5), int(px.mean() + 0.5)
        ox = (px - cx).astype(np.float32)
        oy = (py - cy).astype(np.float32)
        for o in (ox, oy):
            neg, pos = o < 0, o > 0
            if neg.any():
                o[neg] /= -o[neg].min()
            if pos.any():
                o[pos] /= o[pos].max()
        for ch, o in ((0, ox), (1, oy)):
            sub = hv[ch][yw, xw]
            sub[py, px] = o
            hv[ch][yw, xw] = sub
    fg = inst > 0
    hv += np.where(fg, noise, 0.02).astype(np.float32) * rng.standard_normal(hv.shape).astype(np.float32)
    flip = rng.random((H, W)) < 0.10
    tnoise = rng.integers(0, n_types, size=(H, W)).astype(np.uint8)
    tmap = np.where(flip & fg, tnoise, tmap).astype(np.uint8)
    if return_inst:
        return tmap, fg.astype(np.uint8), hv, inst
    return tmap, fg.astype(np.uint8), hv


def world_tile(world, row: int, col: int, patch_size: int = 1024, overlap: int = 64):
    """The (type_map, binary_map, hv) crop a slide tile (row, col) sees: origin = the reference's global tile offset at
    downsampling 1 (cell_detection.py:341-350: row * patch - (row + 0.5) * overlap), wrapped on the torus."""
    tm, bm, hv = world[:3]
    H, W = tm.shape
    y0 = int(row * patch_size - (row + 0.5) * overlap)
    x0 = int(col * patch_size - (col + 0.5) * overlap)
    ys = np.arange(y0, y0 + patch_size) % H
    xs = np.arange(x0, x0 + patch_size) % W
    yw, xw = np.ix_(ys, xs)
    return tm[yw, xw], bm[yw, xw], np.stack([hv[0][yw, xw], hv[1][yw, xw]])
