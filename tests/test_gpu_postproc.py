"""GPU: the on-device HoVer post-processing (C ABI cv_pp_*) is bit-exact against the CPU oracle
(oracle/postproc_ref.c) and the committed golden fixtures on identical input maps."""
import glob
import os

import numpy as np
import pytest
import torch

from cellvit_amd.synth import synth_nuclei_maps

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gpu_chain(tm, bm, hv, mag, nr_types=6, debug=False):
    import ctypes as C
    from cellvit_amd import _lib
    from cellvit_amd.postproc import _PPEngine, _params, postprocess_device, records_to_dicts
    obj, ks = _params(mag)
    dev = torch.device("cuda", 0)
    t = torch.from_numpy(np.ascontiguousarray(tm)).to(dev)
    b = torch.from_numpy(np.ascontiguousarray(bm)).to(dev)
    h = torch.from_numpy(np.ascontiguousarray(hv)).to(dev)
    if t.dim() == 2:
        t, b, h = t[None], b[None], h[None]
    inst, recs, n_recs, contours, n_pts = postprocess_device(b, t, h, nr_types, obj, ks)
    torch.cuda.synchronize()
    dicts = records_to_dicts(recs, n_recs, contours, n_pts)
    taps = None
    if debug:
        B, H, W = b.shape
        e = _PPEngine.get(dev, B, H, W)
        dist = np.empty((B, H, W), np.float64); marker = np.empty((B, H, W), np.int32); blb = np.empty((B, H, W), np.uint8)
        for name, arr in (("dist", dist), ("marker", marker), ("blb", blb)):
            _lib.check(e.lib.cv_pp_debug_read(e.h, name.encode(), arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        taps = (blb, dist, marker)
    return inst.cpu().numpy(), dicts, taps


def _assert_dicts_equal(got, ref):
    assert sorted(got.keys()) == sorted(ref.keys())
    for k in ref:
        assert np.array_equal(got[k]["bbox"], ref[k]["bbox"]), k
        assert np.array_equal(got[k]["centroid"], ref[k]["centroid"]), (k, got[k]["centroid"], ref[k]["centroid"])
        assert got[k]["type"] == ref[k]["type"], k
        assert got[k]["type_prob"] == ref[k]["type_prob"], k
        assert np.array_equal(got[k]["contour"], ref[k]["contour"]), k


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "postproc_*.npz"))))
def test_matches_golden_and_oracle_stage_by_stage(path):
    from oracle import postproc_ref as P
    g = np.load(path)
    idx, size, k, mag = [int(v) for v in g["meta"]]
    tm, bm, hv, _ = synth_nuclei_maps(idx, size, k)
    inst, dicts, (blb, dist, marker) = _gpu_chain(tm, bm, hv, mag, debug=True)
    obj, ks = (10, 21) if mag == 40 else (3, 11)
    o_inst, o_blb, o_dist, o_marker = P.proc_np_hv(bm, hv[0], hv[1], obj, ks, debug=True)
    assert np.array_equal(blb[0], o_blb.astype(np.uint8)), "mask stage (CC + small-object removal)"
    nd = int((dist[0] != o_dist).sum())
    assert nd == 0, f"dist stage (normalise/Sobel/blur): {nd} differing pixels, max |d| {np.abs(dist[0] - o_dist).max()}"
    assert np.array_equal(marker[0], o_marker), "marker stage (fill holes / open / label / size filter)"
    assert np.array_equal(inst[0], o_inst), "watershed"
    assert np.array_equal(inst[0], g["oracle_inst"]) and np.array_equal(inst[0], g["skimage_watershed"])
    d = dicts[0]
    ids = np.array(sorted(d.keys()), dtype=np.int32)
    assert np.array_equal(ids, g["ids"])
    assert np.array_equal(np.array([d[i]["bbox"].ravel() for i in ids]), g["bbox"])
    assert np.array_equal(np.array([d[i]["centroid"] for i in ids]), g["centroid"])
    assert np.array_equal(np.array([d[i]["type"] for i in ids]), g["type"])
    assert np.array_equal(np.array([d[i]["type_prob"] for i in ids]), g["type_prob"])
    assert np.array_equal(np.concatenate([d[i]["contour"] for i in ids]), g["contour_cat"])


@pytest.mark.parametrize("k", [300, 800, 1500])
def test_full_tile_1024_vs_oracle(k):
    """BASELINE.json full tile size, SURVEY §8d cell densities."""
    from oracle import postproc_ref as P
    tm, bm, hv, _ = synth_nuclei_maps(100 + k, 1024, k)
    inst, dicts, _ = _gpu_chain(tm, bm, hv, 40)
    pm = np.stack([tm.astype(np.float32), bm.astype(np.float32), hv[0], hv[1]], -1)
    o_inst, o_d = P.postprocess_tile(pm, 6, 40)
    assert np.array_equal(inst[0], o_inst)
    _assert_dicts_equal(dicts[0], o_d)
    # size-independent properties: instances tile the mask components that own a marker; ids unique per component
    assert ((inst[0] > 0) <= (bm > 0)).all()


def test_batch_of_different_tiles():
    from oracle import postproc_ref as P
    maps = [synth_nuclei_maps(20 + i, 256, 600 + 400 * i) for i in range(3)]
    tm = np.stack([m[0] for m in maps]); bm = np.stack([m[1] for m in maps]); hv = np.stack([m[2] for m in maps])
    inst, dicts, _ = _gpu_chain(tm, bm, hv, 40)
    for i in range(3):
        pm = np.stack([tm[i].astype(np.float32), bm[i].astype(np.float32), hv[i, 0], hv[i, 1]], -1)
        o_inst, o_d = P.postprocess_tile(pm, 6, 40)
        assert np.array_equal(inst[i], o_inst), i
        _assert_dicts_equal(dicts[i], o_d)


def test_batch_of_32_full_tiles_matches_single_tile_runs():
    """bench.py's default batch: 32 tiles of 1024^2 in one launch chain (per-tile bases far apart, one flood queue per
    tile) reproduce the single-tile results bit for bit — first, middle and last tile."""
    m0 = synth_nuclei_maps(40, 1024, 500)
    m1 = synth_nuclei_maps(41, 1024, 900)
    singles = []
    for m in (m0, m1):
        inst, dicts, _ = _gpu_chain(m[0][None], m[1][None], m[2][None], 40)
        singles.append((inst[0].copy(), dicts[0]))
    order = [0] * 32
    for i in (13, 31):
        order[i] = 1
    maps = (m0, m1)
    tm = np.stack([maps[o][0] for o in order]); bm = np.stack([maps[o][1] for o in order]); hv = np.stack([maps[o][2] for o in order])
    inst, dicts, _ = _gpu_chain(tm, bm, hv, 40)
    for i in (0, 13, 30, 31):
        ref_inst, ref_d = singles[order[i]]
        assert np.array_equal(inst[i], ref_inst), i
        _assert_dicts_equal(dicts[i], ref_d)


def test_dense_tile_large_components_overflow_pool():
    """Heavily overlapping nuclei -> few huge mask components: exercises the LDS-pool overflow arena."""
    from oracle import postproc_ref as P
    tm, bm, hv, _ = synth_nuclei_maps(31, 256, 9000)
    assert bm.mean() > 0.6
    inst, dicts, _ = _gpu_chain(tm, bm, hv, 40)
    pm = np.stack([tm.astype(np.float32), bm.astype(np.float32), hv[0], hv[1]], -1)
    o_inst, o_d = P.postprocess_tile(pm, 6, 40)
    assert np.array_equal(inst[0], o_inst)
    _assert_dicts_equal(dicts[0], o_d)


def test_edge_cases_empty_full_and_tiny():
    from oracle import postproc_ref as P
    H = W = 64
    cases = []
    z = np.zeros((H, W), np.uint8)
    cases.append((z, z, np.zeros((2, H, W), np.float32)))                                  # empty tile
    rng = np.random.default_rng(3)
    cases.append((np.full((H, W), 2, np.uint8), np.ones((H, W), np.uint8),
                  rng.standard_normal((2, H, W)).astype(np.float32)))                      # no background pixel
    b = np.zeros((H, W), np.uint8); b[5:8, 5:8] = 1; b[20:40, 20:45] = 1                   # 9-px blob is removed
    cases.append((b * 3, b, (rng.standard_normal((2, H, W)) * 0.1).astype(np.float32)))
    cases.append((b, b, np.zeros((2, H, W), np.float32)))                                  # constant HV: scale 0
    for tm, bm, hv in cases:
        inst, dicts, _ = _gpu_chain(tm, bm, hv, 40)
        pm = np.stack([tm.astype(np.float32), bm.astype(np.float32), hv[0], hv[1]], -1)
        o_inst, o_d = P.postprocess_tile(pm, 6, 40)
        assert np.array_equal(inst[0], o_inst)
        _assert_dicts_equal(dicts[0], o_d)


def test_unknown_magnification_and_reference_class_api():
    from cellvit_amd.postproc import DetectionCellPostProcessor
    from oracle import postproc_ref as P
    with pytest.raises(NotImplementedError):
        DetectionCellPostProcessor(nr_types=6, magnification=10)
    tm, bm, hv, _ = synth_nuclei_maps(40, 256, 700)
    pm = np.stack([tm.astype(np.float32), bm.astype(np.float32), hv[0], hv[1]], -1).astype(np.float64)
    inst, d = DetectionCellPostProcessor(nr_types=6, magnification=20).post_process_cell_segmentation(pm)
    o_inst, o_d = P.postprocess_tile(pm, 6, 20)
    assert np.array_equal(inst, o_inst)
    _assert_dicts_equal(d, o_d)


def test_model_calculate_instance_map_end_to_end():
    """forward (HIP) -> calculate_instance_map (HIP) vs the oracle post-processing of the SAME maps."""
    from cellvit_amd.model import CellViT256
    from oracle import postproc_ref as P
    from helpers import load_case
    cfg, sd, x, _ = load_case("vit256_256")
    m = CellViT256(None, 6, 19, compute_dtype="fp16")
    m.load_state_dict(sd)
    out = m(x.cuda())
    preds = {"nuclei_binary_map": torch.softmax(out["nuclei_binary_map"], 1),
             "nuclei_type_map": torch.softmax(out["nuclei_type_map"], 1), "hv_map": out["hv_map"]}
    inst, dicts = m.calculate_instance_map(preds, magnification=40)
    assert inst.dtype == torch.float32 and tuple(inst.shape) == (1, 256, 256)
    pm = np.concatenate([preds["nuclei_type_map"].argmax(1)[0].cpu().numpy()[..., None],
                         preds["nuclei_binary_map"].argmax(1)[0].cpu().numpy()[..., None],
                         preds["hv_map"][0].permute(1, 2, 0).cpu().numpy()], -1)
    o_inst, o_d = P.postprocess_tile(pm, 6, 40)
    assert np.array_equal(inst[0].numpy().astype(np.int32), o_inst)
    _assert_dicts_equal(dicts[0], o_d)
    with pytest.raises(NotImplementedError):
        m.calculate_instance_map(preds, magnification=30)
