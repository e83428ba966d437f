"""ORACLE (test infrastructure): ctypes wrapper of oracle/postproc_ref.c — the CPU restatement of
post_process_cell_segmentation (/root/reference/cell_segmentation/utils/post_proc_cellvit.py:67-249).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libpostproc_ref.so")


class Instance(C.Structure):
    _fields_ = [("id", C.c_int32), ("rmin", C.c_int32), ("cmin", C.c_int32), ("rmax", C.c_int32),
                ("cmax", C.c_int32), ("npix", C.c_int32), ("type", C.c_int32), ("contour_off", C.c_int32),
                ("contour_len", C.c_int32), ("cx", C.c_double), ("cy", C.c_double), ("type_prob", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "postproc_ref.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        _lib.cvo_label4.restype = C.c_int32
        _lib.cvo_instances.restype = C.c_int
        _lib.cvo_postprocess_tile.restype = C.c_int
    return _lib


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def label4(img: np.ndarray):
    a = np.ascontiguousarray(img.astype(np.int32))
    out = np.empty_like(a)
    n = lib().cvo_label4(_p(a), a.shape[0], a.shape[1], _p(out))
    return out, int(n)


def remove_small(lab: np.ndarray, min_size: int) -> np.ndarray:
    a = np.ascontiguousarray(lab.astype(np.int32)).copy()
    lib().cvo_remove_small(_p(a), a.shape[0], a.shape[1], C.c_int32(int(a.max())), int(min_size))
    return a


def normalize_f32(x: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(a)
    lib().cvo_normalize_f32(_p(a), C.c_int64(a.size), _p(out))
    return out


def normalize_f64(x: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(a.shape, dtype=np.float32)
    lib().cvo_normalize_f64(_p(a), C.c_int64(a.size), _p(out))
    return out


def sobel_kernel(ksize: int, order: int) -> np.ndarray:
    out = np.empty(ksize, dtype=np.float64)
    lib().cvo_sobel_kernel(ksize, order, _p(out))
    return out


def sobel(x: np.ndarray, ksize: int, dx: int) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.float64)
    lib().cvo_sobel(_p(a), a.shape[0], a.shape[1], ksize, dx, _p(out))
    return out


def blur3(x: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(a)
    lib().cvo_blur3(_p(a), a.shape[0], a.shape[1], _p(out))
    return out


def fill_holes(x: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(x.astype(np.uint8))
    out = np.empty_like(a)
    lib().cvo_fill_holes(_p(a), a.shape[0], a.shape[1], _p(out))
    return out


def open5(x: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(x.astype(np.uint8))
    out = np.empty_like(a)
    lib().cvo_open5(_p(a), a.shape[0], a.shape[1], _p(out))
    return out


def watershed(image: np.ndarray, markers: np.ndarray, mask: np.ndarray) -> np.ndarray:
    im = np.ascontiguousarray(image, dtype=np.float64)
    mk = np.ascontiguousarray(markers.astype(np.int32))
    ms = np.ascontiguousarray(mask.astype(np.int32))
    out = np.empty_like(mk)
    lib().cvo_watershed(_p(im), _p(mk), _p(ms), im.shape[0], im.shape[1], _p(out))
    return out


def proc_np_hv(bin_map, hv0, hv1, object_size=10, ksize=21, debug=False):
    b = np.ascontiguousarray(bin_map.astype(np.uint8))
    h0 = np.ascontiguousarray(hv0, dtype=np.float32)
    h1 = np.ascontiguousarray(hv1, dtype=np.float32)
    H, W = b.shape
    inst = np.empty((H, W), dtype=np.int32)
    if debug:
        blb = np.empty((H, W), np.int32); dist = np.empty((H, W), np.float64); marker = np.empty((H, W), np.int32)
    else:
        blb = dist = marker = None
    lib().cvo_proc_np_hv(_p(b), _p(h0), _p(h1), H, W, object_size, ksize, _p(inst), _p(blb), _p(dist), _p(marker))
    return (inst, blb, dist, marker) if debug else inst


def instances(inst: np.ndarray, type_map: np.ndarray, nr_types: int = 6):
    a = np.ascontiguousarray(inst.astype(np.int32))
    t = np.ascontiguousarray(type_map.astype(np.uint8))
    H, W = a.shape
    max_inst = int(a.max()) + 1
    max_pts = 4 * H * W // 8 + 1024
    recs = (Instance * max_inst)()
    pts = np.empty((max_pts, 2), dtype=np.int32)
    npts = C.c_int(0)
    n = lib().cvo_instances(_p(a), _p(t), H, W, nr_types, recs, max_inst, _p(pts), max_pts, C.byref(npts))
    return _records_to_dict(recs, n, pts)


def _records_to_dict(recs, n, pts):
    """Same dict layout as the reference (post_proc:126-151); instances whose simplified contour has
    fewer than 3 points are skipped like the reference does (:113-116)."""
    out = {}
    for i in range(n):
        r = recs[i]
        if r.contour_len < 3:
            continue
        out[int(r.id)] = {
            "bbox": np.array([[r.rmin, r.cmin], [r.rmax, r.cmax]]),
            "centroid": np.array([r.cx, r.cy]),
            "contour": pts[r.contour_off:r.contour_off + r.contour_len].copy(),
            "type_prob": float(r.type_prob),
            "type": int(r.type),
        }
    return out


def postprocess_tile(pred_map: np.ndarray, nr_types: int = 6, magnification: int = 40):
    """pred_map [H, W, 4] = (type argmax, binary argmax, hv0, hv1) as built by
    CellViT.calculate_instance_map (cellvit.py:367-378).  Returns (instance map int32, dict)."""
    if magnification not in (20, 40):
        raise NotImplementedError("Unknown magnification")
    t = np.ascontiguousarray(pred_map[..., 0].astype(np.uint8))
    b = np.ascontiguousarray(pred_map[..., 1].astype(np.uint8))
    h0 = np.ascontiguousarray(pred_map[..., 2], dtype=np.float32)
    h1 = np.ascontiguousarray(pred_map[..., 3], dtype=np.float32)
    H, W = b.shape
    inst = np.empty((H, W), dtype=np.int32)
    max_inst = H * W // 10 + 16
    max_pts = H * W // 2 + 1024
    recs = (Instance * max_inst)()
    pts = np.empty((max_pts, 2), dtype=np.int32)
    npts = C.c_int(0)
    n = lib().cvo_postprocess_tile(_p(t), _p(b), _p(h0), _p(h1), H, W, magnification, nr_types, _p(inst), recs,
                                   max_inst, _p(pts), max_pts, C.byref(npts))
    return inst, _records_to_dict(recs, n, pts)
