"""The native JSON writer (cellvit_amd/csrc/writers.hip, host code: runs without a GPU) on the values its fast paths must not get wrong: negative and 64-bit integers,
doubles that need all 17 digits, integral doubles ("3.0"), exponents, non-finite values as Python's encoder writes them, more cells than one rendering chunk (threads,
chunk seams, the first-cell prefix), contours of length 0 — compared with json.dumps of the same values (reference: json.dump of the per-cell dicts,
cell_segmentation/inference/cell_detection.py:438-457)."""
import ctypes as C
import json
import math

import numpy as np

from cellvit_amd import _lib


def _write(path, n, bbox, cen, ct_off, ct, prob, typ, rc, status, og, edge, ep, det=0, header=b'"wsi_metadata": {"a": 1}'):
    lib = _lib.load()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _lib.check(lib.cv_write_cells_json(str(path).encode(), header, det, n, p(bbox), p(cen), p(ct_off), p(ct), p(prob), p(typ), p(rc), p(status),
                                       p(og), p(edge), p(ep)))
    return open(path).read()


def _arrays(n, rng, L=None):
    lens = rng.integers(0, 7, n) if L is None else np.full(n, L)
    ct_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return dict(bbox=rng.integers(-5, 70000, (n, 4)).astype(np.int64), cen=rng.random((n, 2)) * 65536 - 3.0, ct_off=ct_off,
                ct=rng.integers(-3, 70000, (int(ct_off[-1]) + 1, 2)).astype(np.int64), prob=rng.random(n),
                typ=rng.integers(0, 6, n).astype(np.int32), rc=rng.integers(0, 40, (n, 2)).astype(np.int32),
                status=rng.integers(0, 8, n).astype(np.int32), og=rng.integers(0, 70000, (n, 2)).astype(np.int64),
                edge=(rng.random(n) < 0.3).astype(np.uint8), ep=rng.integers(0, 2, (n, 4)).astype(np.uint8))


def test_numbers_are_written_as_python_writes_them(tmp_path):
    rng = np.random.default_rng(3)
    a = _arrays(12, rng)
    a["bbox"][0] = [-(2 ** 63), 2 ** 63 - 1, -1, 0]
    a["og"][1] = [10 ** 18, -10 ** 18]
    vals = [0.1, 1.0 / 3.0, 3.0, -0.0, 1e-5, 1e22, 5e-324, 1.7976931348623157e308, 123456789012345680.0, 2.5e-7, float("nan"), float("inf")]
    a["prob"][:] = vals
    a["cen"][0] = [float("-inf"), 65535.999999999993]
    doc = json.loads(_write(tmp_path / "c.json", 12, **a))
    cells = doc["cells"]
    assert doc["wsi_metadata"] == {"a": 1} and len(cells) == 12
    assert cells[0]["bbox"] == [[-(2 ** 63), 2 ** 63 - 1], [-1, 0]] and cells[1]["offset_global"] == [10 ** 18, -10 ** 18]
    for k, v in enumerate(vals):
        got = cells[k]["type_prob"]
        assert (math.isnan(got) and math.isnan(v)) or (got == v and math.copysign(1.0, got) == math.copysign(1.0, v)), (k, got, v)
    assert cells[0]["centroid"] == [float("-inf"), 65535.999999999993]
    text = open(tmp_path / "c.json").read()
    assert '"type_prob": 3.0,' in text and '"type_prob": NaN,' in text and '"type_prob": Infinity,' in text and "-Infinity" in text
    for k in range(12):                                                    # contours, incl. empty ones
        assert cells[k]["contour"] == a["ct"][a["ct_off"][k]:a["ct_off"][k + 1]].tolist()


def test_many_cells_across_rendering_chunks_equal_json_dumps(tmp_path):
    rng = np.random.default_rng(5)
    n = 8192 * 3 + 77                                                      # four chunks, the last one short
    a = _arrays(n, rng, L=3)
    for det in (0, 1):
        doc = json.loads(_write(tmp_path / "c.json", n, det=det, **a))
        cells = doc["cells"]
        assert len(cells) == n
        for k in (0, 1, 8191, 8192, 8193, 2 * 8192 - 1, 2 * 8192, 3 * 8192, n - 1):
            c = cells[k]
            assert c["bbox"] == a["bbox"][k].reshape(2, 2).tolist() and c["centroid"] == a["cen"][k].tolist() and c["type"] == int(a["typ"][k])
            if det:
                assert list(c.keys()) == ["bbox", "centroid", "type"]
                continue
            assert c["contour"] == a["ct"][3 * k:3 * k + 3].tolist() and c["type_prob"] == float(a["prob"][k])
            assert c["patch_coordinates"] == a["rc"][k].tolist() and c["cell_status"] == int(a["status"][k]) and c["offset_global"] == a["og"][k].tolist()
            assert c["edge_position"] == bool(a["edge"][k])
            if a["edge"][k]:
                assert c["edge_information"]["position"] == a["ep"][k].tolist()
        if not det:                                                        # key order of the reference's dicts
            assert list(cells[0].keys())[:8] == ["bbox", "centroid", "contour", "type_prob", "type", "patch_coordinates", "cell_status", "offset_global"]
    assert _write(tmp_path / "e.json", 0, **_arrays(0, rng)) == '{"wsi_metadata": {"a": 1}, "cells": []}'
