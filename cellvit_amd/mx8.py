"""OCP MX-fp8 helpers on the host side of the fp8 engine (BASELINE.json configs[4]).

An MX-fp8 tensor is `data` uint8 [rows, K] (OCP e4m3 elements) + one E8M0 scale byte (2^(b - 127)) per 32 consecutive K
elements of a row.  The device keeps the scale bytes in 1-KiB blocks per (256-row tile, 128-element K tile), ordered the way
the MFMA lanes of the 8-phase contraction read them (cellvit_amd/csrc/gemm.h: mx8_scale_off_a / mx8_scale_off_w); the
functions here convert between that order and the plain row-major [rows, K/32] view, and wrap the library's host
quantiser (`cv_mx8_quantize_host`, the one `cv_finalize` applies to the qkv / fc1 / fc2 weights).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

A_SIDE, W_SIDE, ROW_MAJOR = 0, 1, 2


def scale_index(rows: int, K: int, w_side: bool) -> np.ndarray:
    """int64 [rows, K/32]: position of scale (row, k-block) in the tiled device image (restates mx8_scale_index)."""
    r = np.arange(rows, dtype=np.int64)[:, None]
    kb = np.arange(K // 32, dtype=np.int64)[None, :]
    blk = (r >> 8) * (K >> 7) + (kb >> 2)
    rr, g = r & 255, kb & 3
    if w_side:
        off = (((((rr >> 6) * 4 + g) * 16) + (((rr >> 4) & 3) * 4 + (rr & 3))) << 2) | ((rr >> 2) & 3)
    else:
        off = (((((rr >> 7) * 2 + ((rr >> 6) & 1)) * 4 + g) * 16 + (rr & 15)) << 2) | ((rr >> 4) & 3)
    return blk * 1024 + off


def untile_scales(tiled: np.ndarray, rows: int, K: int, w_side: bool) -> np.ndarray:
    """tiled device image (uint8 [rows * K / 32]) -> row-major uint8 [rows, K/32]."""
    return np.asarray(tiled).reshape(-1)[scale_index(rows, K, w_side)]


def tile_scales(row_major: np.ndarray, w_side: bool) -> np.ndarray:
    rows, nb = row_major.shape
    out = np.zeros(rows * nb, np.uint8)
    out[scale_index(rows, nb * 32, w_side).reshape(-1)] = np.asarray(row_major, np.uint8).reshape(-1)
    return out


def quantize(x: np.ndarray, layout: int = ROW_MAJOR):
    """fp32 [rows, K] -> (data uint8 [rows, K], scales uint8 [rows * K / 32] in `layout`) with the library's quantiser."""
    x = np.ascontiguousarray(x, np.float32)
    rows, K = x.shape
    data = np.empty((rows, K), np.uint8)
    sc = np.zeros(rows * (K // 32), np.uint8)
    _lib.check(_lib.load().cv_mx8_quantize_host(x.ctypes.data_as(C.c_void_p), rows, K, int(layout), data.ctypes.data_as(C.c_void_p),
                                                sc.ctypes.data_as(C.c_void_p)))
    return data, sc


_E4M3_LUT = None


def e4m3_to_f32(b: np.ndarray) -> np.ndarray:
    """Decode OCP e4m3 (fn) bytes: 1-4-3, bias 7, subnormals m * 2^-9, 0x7f / 0xff = NaN, max 448."""
    global _E4M3_LUT
    if _E4M3_LUT is None:
        v = np.zeros(256, np.float32)
        for i in range(256):
            s, e, m = i >> 7, (i >> 3) & 15, i & 7
            if e == 15 and m == 7:
                val = np.nan
            elif e == 0:
                val = m * 2.0 ** -9
            else:
                val = (1 + m / 8.0) * 2.0 ** (e - 7)
            v[i] = -val if s else val
        _E4M3_LUT = v
    return _E4M3_LUT[np.asarray(b, np.uint8)]


def dequantize(data: np.ndarray, scales_row_major: np.ndarray) -> np.ndarray:
    """(uint8 [rows, K], uint8 [rows, K/32]) -> float64 [rows, K]."""
    rows, K = data.shape
    s = np.exp2(scales_row_major.astype(np.float64) - 127.0)
    return e4m3_to_f32(data).astype(np.float64) * np.repeat(s, 32, axis=1)
