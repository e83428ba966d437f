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
