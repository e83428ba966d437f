"""CLI: flag surface and host helpers on CPU; an end-to-end run on a tiny synthetic slide on the GPU."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from cellvit_amd.inference import cell_detection as CD


def test_parser_has_the_reference_flags():
    p = CD.InferenceWSIParser()
    a = p.parse_arguments(["--model", "m.pth", "--gpu", "1", "--magnification", "40", "--enforce_amp", "--batch_size", "4",
                           "--outdir_subdir", "x", "--geojson", "process_wsi", "--wsi_path", "a.svs",
                           "--patched_slide_path", "d"])
    assert a["model"] == "m.pth" and a["gpu"] == 1 and a["enforce_amp"] and a["batch_size"] == 4 and a["geojson"]
    assert a["command"] == "process_wsi" and a["wsi_path"] == "a.svs" and a["patched_slide_path"] == "d"
    b = p.parse_arguments(["--model", "m", "process_dataset", "--wsi_paths", "w", "--patch_dataset_path", "p"])
    assert b["command"] == "process_dataset" and b["wsi_extension"] == "svs" and b["filelist"] is None
    with pytest.raises(SystemExit):
        p.parse_arguments(["process_wsi"])       # --model is required


def test_unflatten_dict():
    assert CD.unflatten_dict({"a.b": 1, "a.c.d": 2, "e": 3}) == {"a": {"b": 1, "c": {"d": 2}}, "e": 3}


def _cell(r0, c0, size, row, col, status=None, edge=False, pos=None):
    bbox = np.array([[r0, c0], [r0 + size, c0 + size]])
    cont = np.array([[c0, r0], [c0, r0 + size], [c0 + size, r0 + size], [c0 + size, r0]])
    d = {"bbox": bbox.tolist(), "centroid": [c0 + size / 2, r0 + size / 2], "contour": cont.tolist(), "type": 1,
         "type_prob": 1.0, "patch_coordinates": [row, col], "cell_status": 4 if status is None else status,
         "edge_position": edge}
    if edge:
        d["edge_information"] = {"position": pos, "edge_patches": [[row, col + 1]]}
    return d


def test_stitch_rules():
    cells = [
        _cell(100, 100, 20, 0, 0, status=0),                 # mid cell: always kept
        _cell(500, 970, 20, 0, 0),                           # margin cell of tile (0,0) ...
        _cell(501, 971, 24, 0, 1),                           # ... seen again (larger) by tile (0,1): the larger survives
        _cell(300, 1004, 20, 0, 0, edge=True, pos=[0, 1, 0, 0]),   # edge cell whose neighbour tile exists -> dropped
        _cell(800, 2000, 20, 0, 1, edge=True, pos=[0, 1, 0, 0]),   # edge cell without neighbour tile (0,2) -> kept
    ]
    keep = CD.stitch_cells(cells)
    assert keep == [0, 2, 4]


def test_polygon_geometry_exact():
    sq = np.array([[2, 2], [2, 8], [10, 8], [10, 2]])                       # 8 x 6 rectangle
    assert CD._poly_area(sq) == 48.0
    assert CD._intersection_area(sq, sq + np.array([4, 0])) == 4 * 6       # shifted by 4 in x
    assert CD._intersection_area(sq, sq + np.array([20, 0])) == 0.0
    tri = np.array([[0, 0], [8, 0], [0, 8]])                               # right triangle, area 32
    assert CD._poly_area(tri) == 32.0
    box = np.array([[0, 0], [4, 0], [4, 4], [0, 4]])
    assert abs(CD._intersection_area(tri, box) - 16.0) < 1e-12             # the box lies inside the triangle (x + y <= 8)
    box2 = box + np.array([3, 3])                                          # [3,7]^2 cut by x + y = 8: corner triangle of legs 2
    assert abs(CD._intersection_area(tri, box2) - 2.0) < 1e-12
    ell = np.array([[0, 0], [6, 0], [6, 2], [2, 2], [2, 6], [0, 6]])       # concave L, area 20
    assert CD._poly_area(ell) == 20.0
    assert abs(CD._intersection_area(ell, np.array([[1, 1], [5, 1], [5, 5], [1, 5]])) - (4 + 3 + 0)) < 1e-12
    fa, fb, aa, ab = CD._overlap_fractions({"contour": sq}, {"contour": sq + np.array([4, 0])})
    assert (fa, fb, aa, ab) == (0.5, 0.5, 48.0, 48.0)


@pytest.mark.gpu
def test_cli_end_to_end_tiny_slide(tmp_path):
    from PIL import Image
    from cellvit_amd.spec import cellvit256_config
    from cellvit_amd.weights import make_state_dict, synthetic_tile_u8
    cfg = cellvit256_config()
    ckpt = {"arch": "CellViT256", "model_state_dict": make_state_dict(cfg, 0),
            "config": {"data.num_nuclei_classes": 6, "data.num_tissue_classes": 19, "model.backbone": "default",
                       "training.mixed_precision": True,
                       "dataset_config.nuclei_types": {"Background": 0, "Neoplastic": 1, "Inflammatory": 2,
                                                        "Connective": 3, "Dead": 4, "Epithelial": 5}}}
    torch.save(ckpt, tmp_path / "ckpt.pth")
    slide = tmp_path / "slide"
    (slide / "patches").mkdir(parents=True)
    meta = []
    for col in range(2):
        name = f"slide_0_{col}.png"
        Image.fromarray(synthetic_tile_u8(col, 1024, he_like=True)).save(slide / "patches" / name)
        meta.append({name: {"row": 0, "col": col, "metadata_path": f"metadata/{name}.yaml"}})
    with open(slide / "patch_metadata.json", "w") as f:
        json.dump(meta, f)
    with open(slide / "metadata.yaml", "w") as f:
        yaml.safe_dump({"magnification": 40, "downsampling": 1, "patch_size": 1024, "patch_overlap": 64,
                        "label_map": {"background": 0}, "base_magnification": 40}, f)
    CD.main(["--model", str(tmp_path / "ckpt.pth"), "--batch_size", "2", "--geojson", "process_wsi", "--wsi_path",
             "slide.svs", "--patched_slide_path", str(slide)])
    out = slide / "cell_detection"
    cells = json.load(open(out / "cells.json"))
    assert set(cells.keys()) == {"wsi_metadata", "processed_patches", "type_map", "cells"}
    assert cells["processed_patches"] == ["0_0", "0_1"]
    for c in cells["cells"][:5]:
        assert {"bbox", "centroid", "contour", "type_prob", "type", "patch_coordinates", "cell_status",
                "offset_global", "edge_position"} <= set(c.keys())
    assert (out / "cell_detection.json").exists() and (out / "cells.geojson").exists()
