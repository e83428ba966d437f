"""Wire format of ``cells.pt`` — the cell-graph container the reference CLI pickles with ``torch.save``
(cell_detection.py:469-475): ``CellGraphDataWSI(x, positions, metadata, contours)``, a dataclass defined in
``cell_segmentation/datasets/cell_graph_datamodel.py:18-26`` on top of ``datamodel/graph_datamodel.py:15-29``.

A pickle stores classes by module path, so a file that the reference's own tools can ``torch.load`` must name
``cell_segmentation.datasets.cell_graph_datamodel.CellGraphDataWSI``.  The two dataclasses below carry those module
paths; ``install_reference_aliases()`` registers them under the reference's module names when the reference package
itself is not importable (it never is on the GPU box), so saving and loading work with or without it.
"""
from __future__ import annotations

import sys
import types
from dataclasses import dataclass
from typing import List

import torch

_GRAPH_MOD = "datamodel.graph_datamodel"
_CELL_MOD = "cell_segmentation.datasets.cell_graph_datamodel"


@dataclass
class GraphDataWSI:
    """datamodel/graph_datamodel.py:15-29 — node features, 2-D positions (global WSI coordinates), metadata."""
    x: torch.Tensor
    positions: torch.Tensor
    metadata: dict


@dataclass
class CellGraphDataWSI(GraphDataWSI):
    """cell_graph_datamodel.py:18-26 — plus one contour tensor per cell."""
    contours: List[torch.Tensor]


GraphDataWSI.__module__ = _GRAPH_MOD
CellGraphDataWSI.__module__ = _CELL_MOD


def _ensure_module(name: str) -> types.ModuleType:
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        sub = ".".join(parts[:i])
        if sub not in sys.modules:
            m = types.ModuleType(sub)
            m.__path__ = []          # mark as package so that dotted imports resolve through sys.modules
            sys.modules[sub] = m
            if i > 1:
                setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], m)
    return sys.modules[name]


def install_reference_aliases() -> None:
    """Make ``cell_segmentation.datasets.cell_graph_datamodel.CellGraphDataWSI`` (and its base) resolvable for pickle.
    If the reference package is importable its own classes win and nothing is touched."""
    for mod, cls in ((_GRAPH_MOD, GraphDataWSI), (_CELL_MOD, CellGraphDataWSI)):
        m = sys.modules.get(mod)
        if m is not None and getattr(m, cls.__name__, None) is not None:
            continue
        m = _ensure_module(mod)
        setattr(m, cls.__name__, cls)


def make_cell_graph(x: torch.Tensor, positions: torch.Tensor, contours: List[torch.Tensor], metadata: dict):
    """Build the container with the class pickle will find under the reference's module path."""
    install_reference_aliases()
    cls = getattr(sys.modules[_CELL_MOD], "CellGraphDataWSI")
    return cls(x=x, positions=positions, metadata=metadata, contours=contours)


# ------------------------------------------------------------------------------------------------------------------------
# Fast writer of cells.pt.  `torch.save` pickles a Python list of ~10^6 small tensors at ~30 us per tensor (its pickler calls a
# Python `persistent_id` for every sub-object and `Tensor.__reduce_ex__` per tensor): 11 s per 1024-tile slide, the largest part of
# the slide's writers.  The list entries are views of ONE storage and differ only in (storage_offset, length), so their pickle
# records are generated as one numpy byte array instead: torch.save runs on the same container with a ONE-element contour list
# (so it registers and writes the storages and everything else itself), through `pickle_module=` with a Pickler whose `dump`
# replaces that element's record by the N generated ones.  The file is the same kind of archive with the same kind of pickle
# program (`_rebuild_tensor_v2` on a shared storage, exactly what torch.save emits for views) and loads with plain `torch.load`.
# ------------------------------------------------------------------------------------------------------------------------
def _contour_records(stream: bytes, lengths) -> bytes:
    """Rewrite the pickle of a CellGraphDataWSI whose `contours` list holds ONE [n_points, 2] tensor into the pickle of the same
    object with one [len_k, 2] view per cell."""
    import pickletools

    import numpy as np
    ops = list(pickletools.genops(stream))
    key = next(i for i, (op, arg, pos) in enumerate(ops) if op.name in ("BINUNICODE", "SHORT_BINUNICODE") and arg == "contours")
    i_list = next(i for i in range(key, len(ops)) if ops[i][0].name == "EMPTY_LIST")
    i_app = max(i for i in range(i_list, len(ops)) if ops[i][0].name == "APPEND")
    tail_names = [op.name for op, _, _ in ops[i_app + 1:]]
    if tail_names != ["SETITEMS", "BUILD", "STOP"] or ops[i_list + 1][0].name not in ("BINPUT", "LONG_BINPUT"):
        raise ValueError("unexpected pickle layout")
    ent = ops[i_list + 2:i_app]                     # the single tensor's record
    names = [op.name for op, _, _ in ent]
    if names[0] not in ("BINGET", "LONG_BINGET") or names[1] != "MARK" or "BINPERSID" not in names:
        raise ValueError("unexpected tensor record")
    i_pers = names.index("BINPERSID")
    rebuild_idx = ent[0][1]
    storage_bytes = stream[ent[2][2]:ent[i_pers][2] + 1]          # inner MARK .. BINPERSID: builds the persistent id, loads the storage
    od = [k for k in range(i_pers, len(ent)) if names[k] in ("BINGET", "LONG_BINGET") and names[k + 1] == "EMPTY_TUPLE" and names[k + 2] == "REDUCE"]
    if not od:
        raise ValueError("unexpected tensor record (hooks dict)")
    od_idx = ent[od[0]][1]
    max_put = max([arg for op, arg, _ in ops if op.name in ("BINPUT", "LONG_BINPUT")] + [0])
    S, T = max_put + 1, max_put + 2                                # memo slots of the storage object and of the stride tuple
    lens = np.asarray(lengths, dtype=np.int64)
    n = len(lens)
    if n == 0 or (lens >= 65536).any() or int(lens.sum()) * 2 >= 2 ** 31:
        raise ValueError("contour list outside the fast writer's ranges")
    offs = (np.concatenate([[0], np.cumsum(lens)[:-1]]) * 2).astype("<i4")
    get = lambda idx: (b"h" + bytes([idx])) if idx < 256 else (b"j" + int(idx).to_bytes(4, "little"))   # noqa: E731
    put = lambda idx: b"r" + int(idx).to_bytes(4, "little")                                               # noqa: E731  LONG_BINPUT
    head = get(rebuild_idx) + b"("
    first = (head + storage_bytes + put(S) + b"K\x00" + b"M" + int(lens[0]).to_bytes(2, "little") + b"K\x02\x86" +
             b"K\x02K\x01\x86" + put(T) + b"\x89" + get(od_idx) + b")R" + b"tR")
    body = b""
    if n > 1:
        pre = head + get(S) + b"J"                                   # ... BININT offset
        mid = b"M"                                                   # BININT2 length
        post = b"K\x02\x86" + get(T) + b"\x89" + get(od_idx) + b")R" + b"tR"
        rec = np.zeros((n - 1, len(pre) + 4 + len(mid) + 2 + len(post)), np.uint8)
        c = 0
        rec[:, c:c + len(pre)] = np.frombuffer(pre, np.uint8); c += len(pre)
        rec[:, c:c + 4] = offs[1:].view(np.uint8).reshape(-1, 4); c += 4
        rec[:, c:c + len(mid)] = np.frombuffer(mid, np.uint8); c += len(mid)
        rec[:, c:c + 2] = lens[1:].astype("<u2").view(np.uint8).reshape(-1, 2); c += 2
        rec[:, c:] = np.frombuffer(post, np.uint8)
        body = rec.tobytes()
    return stream[:ent[0][2]] + b"(" + first + body + b"e" + stream[ops[i_app][2] + 1:]


def save_cell_graph(path, x: torch.Tensor, positions: torch.Tensor, contour_points: torch.Tensor, lengths, metadata: dict) -> str:
    """Write cells.pt = CellGraphDataWSI(x, positions, metadata, contours=[points[o_k : o_k + len_k] for k]) (cell_detection.py:469-475).
    contour_points: float32 [sum(lengths), 2].  Returns "fast" or "torch.save" (the route taken)."""
    import io
    import pickle

    pts = contour_points.contiguous()
    lens = [int(v) for v in lengths]
    try:
        if len(lens) < 2 or pts.dtype != torch.float32 or pts.dim() != 2 or pts.shape[1] != 2 or sum(lens) != pts.shape[0]:
            raise ValueError("not the fast writer's case")

        class _Pickler(pickle.Pickler):
            def __init__(self, file, protocol=None, **kw):
                self._out, self._tmp = file, io.BytesIO()
                super().__init__(self._tmp, protocol, **kw)

            def dump(self, obj):
                super().dump(obj)
                self._out.write(_contour_records(self._tmp.getvalue(), lens))

        mod = types.SimpleNamespace(__name__="pickle", Pickler=_Pickler, Unpickler=pickle.Unpickler, dump=pickle.dump,
                                    dumps=pickle.dumps, load=pickle.load, loads=pickle.loads)
        graph = make_cell_graph(x=x, positions=positions, contours=[pts], metadata=metadata)
        torch.save(graph, path, pickle_module=mod, pickle_protocol=2)
        return "fast"
    except Exception:      # anything the byte-stream rewrite can trip over (another torch / pickle layout): the plain route always works
        graph = make_cell_graph(x=x, positions=positions, contours=list(pts.split(lens)), metadata=metadata)
        torch.save(graph, path)
        return "torch.save"
