"""Streaming slide tail: everything the writers need is prepared WHILE the tile loop runs (round 5).

Reference: the slide-level steps run after the whole tile loop — de-duplication, then ``json.dump`` of ~10^6 dicts, then ``torch.save`` of the cell
graph (/root/reference/cell_segmentation/inference/cell_detection.py:423-475).  In this repository's batch route (`cell_detection.finalize_slide` +
`write_outputs`, kept as the checker of this module) that tail was 3.9 s of a 14.5-s slide on one GPU — serial on the writer rank, i.e. the ceiling of
an 8-GPU run.  Here, per finished batch and in a worker thread (ctypes releases the GIL):
  * the batch's token rows are copied device -> pinned staging -> a host array on the copy stream (they used to stay on the device until the end:
    3 GB to copy and to gather per slide);
  * the slide-coordinate geometry of the batch's cells is computed and every cell is rendered ONCE into the two JSON texts (`cv_render_cells`).
After the loop only the margin records are exchanged and de-duplicated (unchanged: `sharding.all_gather_margin_records`, `stitch.stitch_margin_records`);
a keep mask per batch then selects text spans (`cv_textbuf_compact`), token rows, positions and contour points, and the writer rank assembles the
files from chunks in slide order — the batches of all ranks by their first tile index (the block-cyclic shard's blocks ARE the batches):
``cells.json`` / ``cell_detection.json`` byte for byte what `cv_write_cells_json` writes, ``cells.pt`` as a `torch.save` archive written under
`torch.serialization.skip_data` whose tensor holes are filled by `cv_write_rows` (parallel pwrite + CRC-32 for the zip headers).
Ranks other than the writer send their kept chunks point to point, once (text, token rows, positions, contour points: exact sizes).
"""
from __future__ import annotations

import concurrent.futures as cf
import ctypes as C
import json
import struct
import threading
import time
import zipfile
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch

from .. import _lib
from .. import sharding as S


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class _Batch:
    __slots__ = ("key", "ir", "fr", "ct", "n", "tok", "pos32", "cont32", "lens", "tb_cells", "tb_det", "fut", "keep", "geo")


class SlideTail:
    def __init__(self, patch_size: int, downsampling: float, overlap: int, device: Optional[torch.device] = None, workers: int = 2,
                 keep_geometry: bool = False):
        self.ps, self.ds, self.ov = patch_size, downsampling, overlap
        self.device = device
        self.lib = _lib.load()
        self.pool = cf.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="cellvit-tail")
        self.batches: List[_Batch] = []
        self._stage: List[Optional[torch.Tensor]] = [None, None, None]
        self._stage_fut: List[Optional[cf.Future]] = [None, None, None]
        self._turn = 0
        self.keep_geometry = keep_geometry        # the optional geojson pair needs the f64 centroids / i64 contours of the kept cells
        self.token_dim = 0

    # ------------------------------------------------------------------------------------------------ during the tile loop
    def add_batch(self, first_tile: int, ir: np.ndarray, fr: np.ndarray, ct: np.ndarray, tok: Optional[torch.Tensor],
                  copy_stream=None) -> None:
        """Main thread, from the tile loop's `finish`: the packed records of one batch (tiles in ascending order) + their token rows
        [n, D] (device tensor produced on `copy_stream`, or a CPU tensor, or None)."""
        b = _Batch()
        b.key, b.ir, b.fr, b.ct, b.n = int(first_tile), ir, fr, ct, len(ir)
        b.tok = None
        b.keep = None
        stage, ev = None, None
        if tok is not None and b.n:
            self.token_dim = int(tok.shape[1])
            if tok.is_cuda:
                i = self._turn
                self._turn = (self._turn + 1) % len(self._stage)
                if self._stage_fut[i] is not None:
                    self._stage_fut[i].result()                  # the worker has copied this staging buffer out
                st = self._stage[i]
                if st is None or st.shape[0] < b.n or st.shape[1] != tok.shape[1]:
                    st = torch.empty((max(b.n, 1) * 5 // 4, tok.shape[1]), dtype=torch.float32, pin_memory=True)
                    self._stage[i] = st
                stage = st
                with torch.cuda.stream(copy_stream):
                    stage[:b.n].copy_(tok, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                tok.record_stream(copy_stream)
                b.fut = self.pool.submit(self._work, b, stage, ev)
                self._stage_fut[i] = b.fut
                self.batches.append(b)
                return
            b.tok = tok.float().contiguous()
        b.fut = self.pool.submit(self._work, b, None, None)
        self.batches.append(b)

    def _work(self, b: _Batch, stage, ev) -> None:
        from .cell_detection import SlideCells
        if ev is not None:
            ev.synchronize()
            t = torch.empty((b.n, stage.shape[1]), dtype=torch.float32)
            t.copy_(stage[:b.n])
            b.tok = t
        g = SlideCells(b.ir, b.fr, b.ct).geometry(self.ps, self.ds, self.ov)
        typ = np.ascontiguousarray(b.ir[:, S.I_TYPE].astype(np.int32))
        prob = np.ascontiguousarray(b.fr[:, S.F_PROB].astype(np.float64))
        rc = np.ascontiguousarray(b.ir[:, [S.I_ROW, S.I_COL]].astype(np.int32))
        status = np.ascontiguousarray(b.ir[:, S.I_STATUS].astype(np.int32))
        hs = []
        for det in (0, 1):
            h = C.c_void_p()
            _lib.check(self.lib.cv_render_cells(det, b.n, _p(g["bbox"]), _p(g["centroid"]), _p(g["ct_off"]), _p(g["contour"]), _p(prob), _p(typ),
                                                _p(rc), _p(status), _p(g["offset_global"]), _p(g["edge"]), _p(g["edge_pos"]), C.byref(h)))
            hs.append(h)
        b.tb_cells, b.tb_det = hs
        b.pos32 = np.ascontiguousarray(g["centroid"].astype(np.float32))
        b.cont32 = np.ascontiguousarray(g["contour"].astype(np.float32))
        b.lens = np.diff(g["ct_off"]).astype(np.int64)
        b.geo = (g["centroid"], g["ct_off"], g["contour"], typ) if self.keep_geometry else None

    def close(self) -> None:
        # no worker may still be rendering into a batch whose handles are freed below: drop what has not started, wait for what has
        for b in self.batches:
            fut = getattr(b, "fut", None)
            if fut is not None and not fut.cancel():
                try:
                    fut.result()
                except BaseException:      # noqa: BLE001  (the caller is already unwinding, or has seen it via local_margin)
                    pass
        for b in self.batches:
            for h in (getattr(b, "tb_cells", None), getattr(b, "tb_det", None)):
                if h:
                    self.lib.cv_textbuf_free(h)
            b.tb_cells = b.tb_det = None
        self.batches = []
        self.pool.shutdown(wait=False)

    # ------------------------------------------------------------------------------------------------ after the tile loop
    def local_margin(self):
        """The margin records (status != 0) of this rank's batches, in batch order: what the slide-level exchange needs."""
        for b in self.batches:
            b.fut.result()
        irs, frs, cts = [], [], []
        for b in self.batches:
            m = np.nonzero(b.ir[:, S.I_STATUS] != 0)[0]
            lens = b.ir[:, S.I_CLEN].astype(np.int64)
            offs = np.cumsum(lens) - lens
            irs.append(b.ir[m]); frs.append(b.fr[m]); cts.append(S.gather_segments(b.ct.reshape(-1, 2), offs, lens, m))
        if not irs:
            return np.zeros((0, S.N_ICOL), np.int32), np.zeros((0, S.N_FCOL), np.float64), np.zeros((0, 2), np.int32)
        return np.concatenate(irs), np.concatenate(frs), np.concatenate(cts).astype(np.int32).reshape(-1, 2)

    def set_survivors(self, kept_ir: np.ndarray) -> int:
        """kept_ir: the records of the margin cells that survived the global de-duplication.  Builds every batch's keep mask
        (mid cells + surviving margin cells); returns the number of kept local cells."""
        uid = lambda ir: ir[:, S.I_TILE].astype(np.int64) * (1 << 32) + ir[:, S.I_ID].astype(np.int64)   # noqa: E731
        alive = np.unique(uid(kept_ir)) if len(kept_ir) else np.zeros(0, np.int64)
        total = 0
        for b in self.batches:
            keep = (b.ir[:, S.I_STATUS] == 0) | np.isin(uid(b.ir), alive)
            b.keep = np.ascontiguousarray(keep.astype(np.uint8))
            total += int(keep.sum())
        return total

    def _compact_text(self, b: _Batch, which: int) -> np.ndarray:
        h = b.tb_cells if which == 0 else b.tb_det
        need = int(self.lib.cv_textbuf_compact(h, _p(b.keep), None, 0))
        out = np.empty(max(need, 0), np.uint8)
        if need > 0:
            self.lib.cv_textbuf_compact(h, _p(b.keep), _p(out), need)
        return out

    def kept_shard(self, with_arrays: bool = True) -> dict:
        """This rank's kept data as per-batch chunk lists (no concatenation, no copy of the token rows): keys (first tile of the batch),
        counts, the two texts, and (array, keep mask) pairs for token rows / positions / contour points."""
        keys = np.asarray([b.key for b in self.batches], np.int64)
        cnt = np.asarray([int(b.keep.sum()) for b in self.batches], np.int64)
        with cf.ThreadPoolExecutor(max_workers=8) as ex:          # (ctypes calls: the GIL is released while the spans are copied)
            t0 = list(ex.map(lambda b: self._compact_text(b, 0), self.batches))
            t1 = list(ex.map(lambda b: self._compact_text(b, 1), self.batches))
        sh = {"keys": keys, "counts": cnt, "text_cells": t0, "text_det": t1}
        if with_arrays:
            sh["tok"] = [(b.tok, b.keep) for b in self.batches]
            sh["pos"] = [(b.pos32, b.keep) for b in self.batches]
            sh["cont"] = [(b.cont32, np.ascontiguousarray(np.repeat(b.keep, b.lens))) for b in self.batches]
            sh["lens"] = [b.lens[b.keep.astype(bool)] for b in self.batches]
        return sh


# ----------------------------------------------------------------------------------------------------------------------------------
# writer side
# ----------------------------------------------------------------------------------------------------------------------------------
def _zip_data_records(path):
    """name (without the archive prefix) -> (data offset, size, local header offset, central directory entry offset)."""
    out = {}
    with zipfile.ZipFile(path) as z, open(path, "rb") as f:
        cd = z.start_dir
        for i in z.infolist():
            f.seek(i.header_offset)
            h = f.read(30)
            nlen, elen = struct.unpack("<HH", h[26:30])
            f.seek(cd)
            ch = f.read(46)
            cn, ce, cc = struct.unpack("<HHH", ch[28:34])
            out[i.filename.split("/", 1)[1]] = (i.header_offset + 30 + nlen + elen, i.file_size, i.header_offset, cd)
            cd += 46 + cn + ce + cc
    return out


def _zip_patch_crc(path, recs, crcs):
    with open(path, "r+b") as f:
        for name, crc in crcs.items():
            _, _, lho, cdo = recs[name]
            f.seek(lho + 14); f.write(struct.pack("<I", crc & 0xFFFFFFFF))
            f.seek(cdo + 16); f.write(struct.pack("<I", crc & 0xFFFFFFFF))


def _write_rows(lib, path, offset, row_bytes, chunks):
    """chunks: list of (host array with rows of row_bytes, keep mask uint8 or None).  Returns (crc32, rows written)."""
    chunks = [(a, k) for a, k in chunks if a is not None and len(a)]
    n = len(chunks)
    ptrs = (C.c_void_p * max(n, 1))()
    rows = (C.c_int64 * max(n, 1))()
    keeps = (C.c_void_p * max(n, 1))()
    hold = []
    for i, (a, k) in enumerate(chunks):
        if isinstance(a, torch.Tensor):
            a = a.contiguous(); ptrs[i] = a.data_ptr(); rows[i] = a.shape[0]
        else:
            a = np.ascontiguousarray(a); ptrs[i] = a.ctypes.data; rows[i] = a.shape[0]
        if k is not None:
            k = np.ascontiguousarray(k.astype(np.uint8, copy=False)); keeps[i] = k.ctypes.data
        else:
            keeps[i] = None
        hold.append((a, k))
    crc, nrows = C.c_uint32(0), C.c_int64(0)
    _lib.check(lib.cv_write_rows(str(path).encode(), int(offset), int(row_bytes), n, ptrs, rows, keeps, C.byref(crc), C.byref(nrows)))
    return int(crc.value), int(nrows.value)


def write_cells_pt_streamed(path, n_kept: int, D: int, tok_chunks, pos_chunks, cont_chunks, lens: np.ndarray, metadata: dict) -> str:
    """cells.pt = CellGraphDataWSI(x [n, D], positions [n, 2], metadata, contours = one [len_k, 2] view per cell) (cell_detection.py:469-475),
    written as: torch.save of the container under skip_data (pickle + zip headers, holes for the three storages), the holes filled by
    cv_write_rows from the chunk lists, the CRC-32 fields patched afterwards.  On any surprise (no skip_data in this torch, an archive
    layout other than the expected one, a short row write) the kept chunks are concatenated and the file is rewritten by the plain
    torch.save route of datamodel.save_cell_graph — the slide keeps its cells.pt either way."""
    from ..datamodel import save_cell_graph
    m = int(lens.sum())
    try:
        return _write_cells_pt_holes(path, n_kept, D, tok_chunks, pos_chunks, cont_chunks, lens, metadata, m)
    except Exception as e:      # noqa: BLE001
        import logging
        logging.getLogger("cellvit_amd").warning(f"cells.pt: streamed write failed ({e!r}); rewriting through torch.save")

    def kept(chunks, width):
        rows = []
        for a, k in chunks:
            if a is None or not len(a):
                continue
            t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
            rows.append(t if k is None else t[torch.from_numpy(np.ascontiguousarray(k).astype(bool))])
        return torch.cat(rows).float().reshape(-1, width) if rows else torch.zeros((0, width), dtype=torch.float32)
    return save_cell_graph(path, kept(tok_chunks, D), kept(pos_chunks, 2), kept(cont_chunks, 2), lens.tolist(), metadata)


def _write_cells_pt_holes(path, n_kept, D, tok_chunks, pos_chunks, cont_chunks, lens, metadata, m):
    from torch.serialization import skip_data
    from ..datamodel import save_cell_graph
    lib = _lib.load()
    if max(n_kept * D * 4, m * 8) >= (1 << 32) - (1 << 20):
        # a record of 4 GiB or more makes the archive writer switch to zip64 headers / data descriptors, a layout the hole offsets and CRC patch
        # positions below have never been checked against: such slides (≈ 1.4e6 SAM-H cells) take the plain torch.save route
        raise RuntimeError("a cells.pt record of 4 GiB or more: zip64 layout, not filled in place")
    with skip_data():
        route = save_cell_graph(path, torch.empty((n_kept, D), dtype=torch.float32), torch.empty((n_kept, 2), dtype=torch.float32),
                                torch.empty((m, 2), dtype=torch.float32), lens.tolist(), metadata)
    recs = _zip_data_records(path)
    want = {"data/0": (n_kept * D * 4, D * 4, tok_chunks), "data/1": (n_kept * 8, 8, pos_chunks), "data/2": (m * 8, 8, cont_chunks)}
    crcs = {}
    for name, (size, row_bytes, chunks) in want.items():
        if size == 0:
            continue
        if name not in recs or recs[name][1] != size:
            raise RuntimeError(f"cells.pt: unexpected archive layout ({name}: {recs.get(name)}, wanted {size} bytes)")
        crc, nrows = _write_rows(lib, path, recs[name][0], row_bytes, chunks)
        if nrows * row_bytes != size:
            raise RuntimeError(f"cells.pt: {name} got {nrows} rows of {row_bytes} bytes, wanted {size} bytes")
        crcs[name] = crc
    _zip_patch_crc(path, recs, crcs)
    return route


def write_json_chunks(path, header: bytes, chunks: List[np.ndarray]) -> None:
    """`{header, "cells": [` + the non-empty chunks joined by ",\\n" + `]}` — the document of cv_write_cells_json."""
    nz = [c for c in chunks if len(c)]
    # buffered writer: BufferedWriter.write() takes all bytes or raises (a raw FileIO write may be short, e.g. before ENOSPC or beyond
    # 0x7ffff000 bytes per call); the flush inside the with-block surfaces the OS error here, not at garbage collection
    with open(path, "wb", buffering=1 << 20) as f:
        f.write(b"{" + header + b", \"cells\": [")
        for i, c in enumerate(nz):
            f.write(b"\n" if i == 0 else b",\n")
            mv = memoryview(c).cast("B")
            for o in range(0, len(mv), 1 << 30):
                f.write(mv[o:o + (1 << 30)])
        f.write(b"\n]}" if nz else b"]}")
        f.flush()
