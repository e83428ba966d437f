"""Host-side mirror of the reference post-processing interface, backed by the on-device HIP chain.

Reference surface reproduced (same names / argument meaning / error behaviour):
  * ``DetectionCellPostProcessor(nr_types, magnification, gt).post_process_cell_segmentation(pred_map)``
      — cell_segmentation/utils/post_proc_cellvit.py:33-153
  * ``calculate_instance_map(predictions, magnification)`` glue of CellViT — cellvit.py:332-383
  * ``calculate_instances(pred_types, pred_insts)`` (ground-truth side of the evaluation callers) — post_proc_cellvit.py:252-330
The numerical work (CC labelling, Sobel, marker watershed, per-instance records, contours) runs in
libcellvit_amd.so on the GPU; this module only moves pointers and unpacks the record arrays.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Literal, Optional, Tuple

import numpy as np
import torch

from . import _lib


# Capacity of the per-tile device arrays, as divisors of the pixel count.  The defaults hold 8192 instance records and
# 131072 contour points per 1024^2 tile (a dense tile has ~1500 nuclei).  A tile that exceeds them is REPORTED
# (records_to_dicts raises CapacityError); call set_capacity() with smaller divisors and re-run.
_CAP = {"inst_div": 128, "pts_div": 8}


class CapacityError(RuntimeError):
    """More instances / contour points in a tile than the post-processing handle has slots for."""


def set_capacity(inst_div: int = 128, pts_div: int = 8) -> None:
    """Record slots per tile = H*W // inst_div, contour points per tile = H*W // pts_div (handles are rebuilt lazily)."""
    _CAP["inst_div"], _CAP["pts_div"] = int(inst_div), int(pts_div)
    for e in list(_PPEngine._cache.values()):
        e.close()
    _PPEngine._cache.clear()


class _PPEngine:
    _cache: Dict[tuple, "_PPEngine"] = {}

    def __init__(self, device: torch.device, B: int, H: int, W: int):
        self.lib = _lib.load()
        self.B, self.H, self.W = B, H, W
        self.max_inst = max(256, H * W // _CAP["inst_div"])   # record slots per tile (1024^2 -> 8192)
        self.max_pts = max(4096, H * W // _CAP["pts_div"])    # contour points per tile
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.cv_pp_create(B, H, W, self.max_inst, self.max_pts, C.byref(h)))
        self.h = h
        self.device = device

    @classmethod
    def get(cls, device: torch.device, B: int, H: int, W: int) -> "_PPEngine":
        key = (device.index if device.index is not None else torch.cuda.current_device(), H, W)
        e = cls._cache.get(key)
        if e is None or e.B < B:
            if e is not None:
                e.close()
            e = cls(device, B, H, W)
            cls._cache[key] = e
        return e

    def close(self):
        if getattr(self, "h", None):
            self.lib.cv_pp_destroy(self.h)
            self.h = None


def postprocess_device(bin_argmax: torch.Tensor, type_argmax: Optional[torch.Tensor], hv: torch.Tensor,
                       nr_types: int, object_size: int, ksize: int, want_contours: bool = True):
    """Run the chain for a batch that is already on the GPU.  Returns device tensors
    (inst_map i32 [B,H,W], recs u8 [B,max_inst,sizeof(cv_instance)], n_recs i32 [B],
    contours i32 [B,max_pts,2] | None, n_pts i32 [B]) — nothing is copied to the host."""
    if not bin_argmax.is_cuda:
        raise RuntimeError("cellvit_amd post-processing runs on the MI355X only (no CPU fallback)")
    dev = bin_argmax.device
    B, H, W = bin_argmax.shape
    e = _PPEngine.get(dev, B, H, W)
    bin_argmax = bin_argmax.contiguous()
    hv = hv.contiguous().float()
    if type_argmax is not None:
        type_argmax = type_argmax.contiguous()
    with torch.cuda.device(dev):
        inst = torch.empty((B, H, W), device=dev, dtype=torch.int32)
        recs = torch.empty((B, e.max_inst, C.sizeof(_lib.cv_instance)), device=dev, dtype=torch.uint8)
        n_recs = torch.zeros((B,), device=dev, dtype=torch.int32)
        n_pts = torch.zeros((B,), device=dev, dtype=torch.int32)
        contours = torch.empty((B, e.max_pts, 2), device=dev, dtype=torch.int32) if want_contours else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(e.lib.cv_pp_run_params(
            e.h, bin_argmax.data_ptr(), type_argmax.data_ptr() if type_argmax is not None else None, hv.data_ptr(), B,
            int(object_size), int(ksize), int(nr_types if type_argmax is not None else 0), inst.data_ptr(),
            recs.data_ptr(), n_recs.data_ptr(), contours.data_ptr() if contours is not None else None,
            n_pts.data_ptr(), C.c_void_p(stream)))
    return inst, recs, n_recs, contours, n_pts


def check_capacity(recs: torch.Tensor, n_recs_host: np.ndarray, contours: Optional[torch.Tensor],
                   n_pts_host: np.ndarray) -> None:
    """cv_pp_run reports the TRUE per-tile counts; anything beyond the handle's capacity was not written."""
    max_inst = recs.shape[1]
    if (n_recs_host > max_inst).any():
        b = int(np.argmax(n_recs_host))
        raise CapacityError(f"tile {b} of the batch has {int(n_recs_host[b])} instances but the post-processing handle holds "
                            f"{max_inst} records per tile: cellvit_amd.postproc.set_capacity(inst_div=...) and re-run")
    if contours is not None and (n_pts_host > contours.shape[1]).any():
        b = int(np.argmax(n_pts_host))
        raise CapacityError(f"tile {b} of the batch needs {int(n_pts_host[b])} contour points but the post-processing handle "
                            f"holds {contours.shape[1]} per tile: cellvit_amd.postproc.set_capacity(pts_div=...) and re-run")


def records_to_dicts(recs: torch.Tensor, n_recs: torch.Tensor, contours: Optional[torch.Tensor],
                     n_pts: torch.Tensor, return_index: bool = False):
    """Device record arrays -> the reference's per-tile dicts (post_proc:126-151).  This is the point
    where cell records leave the device (the writer / geojson step).  With ``return_index`` also returns, per tile,
    the record slot of every dict entry (in dict order) — the row of that cell in cv_pool_tokens' output."""
    nr = n_recs.cpu().numpy()
    npt = n_pts.cpu().numpy()
    check_capacity(recs, nr, contours, npt)
    out, index = [], []
    for b in range(recs.shape[0]):
        n = int(nr[b])
        raw = recs[b, :n].cpu().numpy().tobytes()
        arr = (_lib.cv_instance * n).from_buffer_copy(raw) if n else []
        pts = contours[b, : int(npt[b])].cpu().numpy() if contours is not None else None
        d, idx = {}, []
        for slot, r in enumerate(arr):
            if pts is not None and r.contour_len < 3:      # "< 3 points dont make a contour" (post_proc:113-116)
                continue
            d[int(r.id)] = {
                "bbox": np.array([[r.rmin, r.cmin], [r.rmax, r.cmax]]),
                "centroid": np.array([r.cx, r.cy]),
                "contour": pts[r.contour_off:r.contour_off + r.contour_len].copy() if pts is not None else None,
                "type_prob": float(r.type_prob),
                "type": int(r.type),
            }
            idx.append(slot)
        out.append(d)
        index.append(idx)
    return (out, index) if return_index else out


def argmax_channels(x: torch.Tensor) -> torch.Tensor:
    """torch.argmax(x, dim=1) of an fp32 BCHW map as a HIP kernel (first maximum) -> u8 [B,H,W] on the device."""
    if not x.is_cuda:
        raise RuntimeError("cellvit_amd runs on the MI355X only (no CPU fallback)")
    x = x.contiguous().float()
    B, Cn, H, W = x.shape
    out = torch.empty((B, H, W), device=x.device, dtype=torch.uint8)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cv_op_argmax_nchw(x.data_ptr(), out.data_ptr(), B, Cn, H, W,
                                                 C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return out


def pool_cell_tokens(tokens: torch.Tensor, recs: torch.Tensor, n_recs: torch.Tensor, patch_size: int = 16
                     ) -> Tuple[torch.Tensor, np.ndarray]:
    """Cell-token pooling of the inference CLI (cell_detection.py:396-409) for every instance record of a batch, on
    the device: tokens [B, D, gh, gw] (the `tokens` entry of forward) -> (fp32 [sum n_recs, D] device tensor, host
    int64 [B] row offset of each tile).  One kernel launch, no per-cell host round trip."""
    B, D, gh, gw = tokens.shape
    tok_nhwc = tokens.permute(0, 2, 3, 1).contiguous().float()       # a view of forward's NHWC buffer: no copy
    nr = n_recs.cpu().numpy().astype(np.int64)
    max_inst = recs.shape[1]
    nr = np.minimum(nr, max_inst)
    off = np.concatenate([[0], np.cumsum(nr)[:-1]]).astype(np.int64)
    total, max_n = int(nr.sum()), int(nr.max()) if len(nr) else 0
    out = torch.empty((total, D), device=tokens.device, dtype=torch.float32)
    if total:
        off_dev = torch.from_numpy(off).to(tokens.device)
        with torch.cuda.device(tokens.device):
            _lib.check(_lib.load().cv_pool_tokens(tok_nhwc.data_ptr(), B, gh, gw, D, int(patch_size), recs.data_ptr(),
                                                  max_inst, n_recs.data_ptr(), off_dev.data_ptr(), max_n, out.data_ptr(),
                                                  C.c_void_p(torch.cuda.current_stream(tokens.device).cuda_stream)))
    return out, off


def _params(magnification: int, gt: bool = False) -> Tuple[int, int]:
    if magnification == 40:
        object_size, k_size = 10, 21
    elif magnification == 20:
        object_size, k_size = 3, 11
    else:
        raise NotImplementedError("Unknown magnification")
    if gt:
        object_size, k_size = 100, 21
    return object_size, k_size


class DetectionCellPostProcessor:
    """post_proc_cellvit.py:33-65 — same constructor, same ``post_process_cell_segmentation``."""

    def __init__(self, nr_types: int = None, magnification: Literal[20, 40] = 40, gt: bool = False) -> None:
        self.nr_types = nr_types
        self.magnification = magnification
        self.gt = gt
        self.object_size, self.k_size = _params(magnification, gt)

    def post_process_cell_segmentation(self, pred_map: np.ndarray, device: Optional[torch.device] = None
                                       ) -> Tuple[np.ndarray, dict]:
        """pred_map [H, W, 4] = (type, binary, hv0, hv1) (or [H, W, 3] without types) -> (instance map, dict)."""
        device = device or torch.device("cuda", torch.cuda.current_device())
        pm = torch.as_tensor(np.ascontiguousarray(pred_map))
        if self.nr_types is not None:
            typ = pm[..., 0].to(torch.int32).to(torch.uint8)[None].to(device)
            inst_ch = pm[..., 1:]
        else:
            typ, inst_ch = None, pm
        binm = (inst_ch[..., 0].float() >= 0.5).to(torch.uint8)[None].to(device)
        hv = inst_ch[..., 1:3].float().permute(2, 0, 1)[None].to(device)
        inst, recs, n_recs, contours, n_pts = postprocess_device(binm, typ, hv, self.nr_types or 0, self.object_size,
                                                                 self.k_size)
        d = records_to_dicts(recs, n_recs, contours, n_pts)[0]
        return inst[0].cpu().numpy(), d


def calculate_instance_map(predictions: dict, num_nuclei_classes: int, magnification: Literal[20, 40] = 40,
                           argmax_planes: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
                           ) -> Tuple[torch.Tensor, List[dict]]:
    """cellvit.py:332-383: predictions (BCHW, softmax or raw logits — only the argmax is consumed) ->
    (instance maps [B,H,W] float32 on the host like the reference, list of per-image nucleus dicts).
    ``argmax_planes`` = (binary, type) u8 planes written by the forward kernels for exactly these maps; without them
    the channel argmax (cellvit.py:366-374) runs as a HIP kernel on the given maps."""
    object_size, k_size = _params(magnification)
    if argmax_planes is not None:
        binm, typ = argmax_planes
    else:
        binm = argmax_channels(predictions["nuclei_binary_map"])
        typ = argmax_channels(predictions["nuclei_type_map"])
    hv = predictions["hv_map"]
    inst, recs, n_recs, contours, n_pts = postprocess_device(binm, typ, hv, num_nuclei_classes, object_size, k_size)
    return inst.float().cpu(), records_to_dicts(recs, n_recs, contours, n_pts)


def calculate_instances(pred_types: torch.Tensor, pred_insts: torch.Tensor) -> List[dict]:
    """post_proc_cellvit.py:252-330 ("best used for GT"): (one-hot / score type map [B, C, H, W], instance map [B, H, W])
    -> per image {id: {bbox, centroid, contour, type_prob, type}} with the reference's conventions (np.unique()[1:] ids,
    majority type with background replaced by the runner-up, instances whose contour has < 3 points dropped).
    The per-instance statistics and the contour tracing are the P7/P8 kernels of the prediction path, run on the given
    ids (`cv_pp_records`); only the record arrays come back to the host."""
    if not pred_insts.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("cellvit_amd post-processing runs on the MI355X only (no CPU fallback)")
        pred_insts = pred_insts.cuda()
    dev = pred_insts.device
    B, H, W = pred_insts.shape
    typ = argmax_channels(pred_types.to(dev)) if pred_types.shape[1] > 1 else torch.zeros((B, H, W), device=dev, dtype=torch.uint8)
    nr_types = int(pred_types.shape[1])
    if nr_types > 256:    # u8 type planes; more than 8 classes vote in windows of 8 on the device (post_proc_cellvit.py:300-318: np.unique has no limit)
        raise NotImplementedError(f"calculate_instances: {nr_types} nucleus classes, the type planes are uint8")
    e = _PPEngine.get(dev, B, H, W)
    if int(pred_insts.max()) > H * W // 16:
        raise CapacityError(f"instance ids above {H * W // 16} have no accumulator slot on a {H}x{W} tile: remap the labels first")
    with torch.cuda.device(dev):
        inst = pred_insts.to(torch.int32).contiguous().clone()
        recs = torch.empty((B, e.max_inst, C.sizeof(_lib.cv_instance)), device=dev, dtype=torch.uint8)
        n_recs = torch.zeros((B,), device=dev, dtype=torch.int32)
        n_pts = torch.zeros((B,), device=dev, dtype=torch.int32)
        contours = torch.empty((B, e.max_pts, 2), device=dev, dtype=torch.int32)
        _lib.check(e.lib.cv_pp_records(e.h, inst.data_ptr(), typ.data_ptr(), B, nr_types, recs.data_ptr(),
                                       n_recs.data_ptr(), contours.data_ptr(), n_pts.data_ptr(),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return records_to_dicts(recs, n_recs, contours, n_pts)
