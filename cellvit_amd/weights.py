"""Deterministic synthetic weights for the reference checkpoint layout.

Trained CellViT checkpoints are Google-Drive links (reference README.md:213-216) and are not
available offline, so parity and benchmarks run on *seeded random* weights.  Torch's default
initialisation is unsuitable: SAM ``pos_embed`` and every ``rel_pos_*`` are zero-initialised
(SAM/image_encoder.py:75-79, 232-233) and BatchNorm running stats are 0/1, which would leave
those code paths untested.  This generator therefore randomises *every* tensor, keyed only by
the ordered (key, shape, kind) list of :mod:`cellvit_amd.spec` — the GPU box regenerates the
identical state_dict from the seed, no reference code needed.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from .spec import CellViTConfig, param_specs


def _draw(rng: np.random.Generator, shape, kind: str) -> np.ndarray:
    if kind == "w":  # Linear / Conv2d [out, in, ...]: N(0, 1/fan_in)
        fan_in = int(np.prod(shape[1:]))
        return rng.standard_normal(shape) / np.sqrt(fan_in)
    if kind == "wt":  # ConvTranspose2d [in, out, 2, 2]: each output pixel sees `in` taps
        return rng.standard_normal(shape) / np.sqrt(shape[0])
    if kind in ("b", "ln_b", "bn_b", "bn_mean"):
        return 0.1 * rng.standard_normal(shape)
    if kind in ("ln_w", "bn_w", "bn_var"):
        return rng.uniform(0.5, 1.5, size=shape)
    if kind == "emb":
        return 0.02 * rng.standard_normal(shape)
    raise ValueError(kind)


def make_state_dict_numpy(cfg: CellViTConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """One draw per tensor, in state_dict order, from ``numpy.random.default_rng(seed)``."""
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape, kind in param_specs(cfg):
        if kind == "bn_count":
            sd[key] = np.zeros((), dtype=np.int64)
            continue
        std_boost = 5.0 if ".rel_pos_" in key else 1.0  # make the rel-pos bias path visible
        sd[key] = (std_boost * _draw(rng, shape, kind)).astype(np.float32)
    return sd


def make_state_dict(cfg: CellViTConfig, seed: int = 0):
    import torch

    return OrderedDict((k, torch.from_numpy(v.copy()) if v.ndim else torch.tensor(int(v)))
                       for k, v in make_state_dict_numpy(cfg, seed).items())


def synthetic_tile_u8(tile_idx: int, size: int = 1024, he_like: bool = False) -> np.ndarray:
    """Seeded uint8 RGB tile [size, size, 3] (SURVEY §8d 'Synthetic inputs — forward')."""
    if not he_like:
        rng = np.random.default_rng(1234 + tile_idx)
        return rng.integers(0, 256, size=(size, size, 3), dtype=np.uint8)
    rng = np.random.default_rng(42 + tile_idx)
    img = np.empty((size, size, 3), dtype=np.float32)
    img[:] = np.array([225, 190, 215], dtype=np.float32)
    img += rng.uniform(-8, 8, size=img.shape).astype(np.float32)
    n = int(rng.integers(400, 801) * (size / 1024.0) ** 2) + 1
    yy, xx = np.mgrid[0:size, 0:size]
    for _ in range(n):
        cy, cx = rng.uniform(0, size, 2)
        ra, rb = rng.uniform(8, 18, 2)
        th = rng.uniform(0, np.pi)
        y0, y1 = int(max(cy - 20, 0)), int(min(cy + 20, size))
        x0, x1 = int(max(cx - 20, 0)), int(min(cx + 20, size))
        if y0 >= y1 or x0 >= x1:
            continue
        dy = yy[y0:y1, x0:x1] - cy
        dx = xx[y0:y1, x0:x1] - cx
        u = dx * np.cos(th) + dy * np.sin(th)
        v = -dx * np.sin(th) + dy * np.cos(th)
        m = (u / ra) ** 2 + (v / rb) ** 2 <= 1.0
        img[y0:y1, x0:x1][m] = np.array([90, 50, 130], dtype=np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def normalize_tile(tile_u8: np.ndarray, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)) -> np.ndarray:
    """ToTensor + Normalize of the reference CLI (cell_detection.py:214-227): HWC u8 → CHW f32."""
    x = tile_u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1)).astype(np.float32)
