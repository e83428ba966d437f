"""CPU: pin the oracle's C restatement of the post-processing.

* scipy primitives (label, binary_fill_holes): black-box comparison with the installed scipy.
* skimage watershed: golden vectors produced by skimage itself (tools/make_golden_postproc.py).
* OpenCV-derived stages: known-answer tests of the documented semantics (parity unpinned, no cv2 here)
  + an independent numpy/scipy composition of the whole __proc_np_hv chain.
"""
import glob
import os

import numpy as np
import pytest
from scipy import ndimage

from cellvit_amd.synth import synth_nuclei_maps
from oracle import postproc_ref as P

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("seed", range(4))
def test_label4_matches_scipy(seed):
    rng = np.random.default_rng(seed)
    img = (rng.random((97, 131)) < (0.35 + 0.1 * seed)).astype(np.int32)
    ref, n = ndimage.label(img)
    got, m = P.label4(img)
    assert n == m and (ref == got).all()


def test_label4_edge_cases():
    for img in (np.zeros((5, 7), np.int32), np.ones((5, 7), np.int32), np.eye(6, dtype=np.int32)):
        ref, n = ndimage.label(img)
        got, m = P.label4(img)
        assert n == m and (ref == got).all()


@pytest.mark.parametrize("seed", range(3))
def test_fill_holes_matches_scipy(seed):
    rng = np.random.default_rng(10 + seed)
    img = ndimage.binary_dilation(rng.random((80, 90)) < 0.08, iterations=2)
    assert (P.fill_holes(img) == ndimage.binary_fill_holes(img)).all()
    ring = np.zeros((9, 9), np.uint8); ring[1:8, 1:8] = 1; ring[3:6, 3:6] = 0; ring[4, 0:4] = 0   # hole open to border
    assert (P.fill_holes(ring) == ndimage.binary_fill_holes(ring)).all()


def test_remove_small_keeps_ids():
    lab = np.zeros((6, 12), np.int32)
    lab[0, :3] = 1; lab[2:5, 2:6] = 2; lab[5, 9:11] = 3
    out = P.remove_small(lab, 10)
    assert set(np.unique(out)) == {0, 2}          # tools.py:61-101: no relabel, ids survive with gaps


def test_sobel_kernels_known_answers():
    assert P.sobel_kernel(3, 1).tolist() == [-1, 0, 1]
    assert P.sobel_kernel(3, 0).tolist() == [1, 2, 1]
    assert P.sobel_kernel(5, 1).tolist() == [-1, -2, 0, 2, 1]
    assert P.sobel_kernel(5, 0).tolist() == [1, 4, 6, 4, 1]
    from math import comb
    assert P.sobel_kernel(21, 0).tolist() == [comb(20, j) for j in range(21)]
    d = np.convolve([comb(19, j) for j in range(20)], [1, -1])[::-1] * -1   # binomial(19) * [-1, +1]
    k = P.sobel_kernel(21, 1)
    assert k.tolist() == np.convolve([comb(19, j) for j in range(20)], [-1, 1]).tolist()
    assert k.sum() == 0 and (k[:10] < 0).all() and (k[11:] > 0).all()


def test_sobel_impulse_and_ramp():
    img = np.zeros((61, 61), np.float32); img[30, 30] = 1
    sm, dv = P.sobel_kernel(21, 0), P.sobel_kernel(21, 1)
    gx = P.sobel(img, 21, 1)
    # correlation: response at (y, x) = k[(30 - x) + 10] -> flipped kernel around the impulse
    assert np.array_equal(gx[20:41, 20:41], np.outer(sm, dv[::-1]))
    gy = P.sobel(img, 21, 0)
    assert np.array_equal(gy[20:41, 20:41], np.outer(dv[::-1], sm))
    ramp = np.tile(np.arange(64, dtype=np.float32), (64, 1))            # increases left -> right
    g = P.sobel(ramp, 21, 1)
    interior = g[:, 10:54]
    assert (interior > 0).all() and np.allclose(interior, interior[0, 0])
    assert np.allclose(P.sobel(ramp, 21, 0)[10:54], 0)
    assert np.allclose(P.sobel(ramp.T.copy(), 21, 0)[10:54, :], interior[0, 0])


def test_normalize_minmax():
    x = np.array([[2.0, 4.0, 3.0]], np.float32)
    assert P.normalize_f32(x).tolist() == [[0.0, 1.0, 0.5]]
    assert P.normalize_f64(x.astype(np.float64)).tolist() == [[0.0, 1.0, 0.5]]
    c = np.full((3, 3), 7.5, np.float32)                                 # constant: scale 0 -> all shift (= 0)
    assert (P.normalize_f32(c) == 0).all() and (P.normalize_f64(c.astype(np.float64)) == 0).all()
    rng = np.random.default_rng(0)
    y = rng.standard_normal((50, 50)).astype(np.float32)
    n = P.normalize_f32(y)
    assert n.dtype == np.float32 and abs(n.min()) < 1e-6 and abs(n.max() - 1) < 1e-6
    assert np.allclose(n, (y - y.min()) / (y.max() - y.min()), atol=2e-7)


def test_blur3_impulse_and_border():
    img = np.zeros((7, 7)); img[3, 3] = 1
    k = np.array([0.25, 0.5, 0.25])
    assert np.array_equal(P.blur3(img)[2:5, 2:5], np.outer(k, k))
    assert np.array_equal(P.blur3(np.ones((5, 6))), np.ones((5, 6)))
    corner = np.zeros((5, 5)); corner[0, 0] = 1                         # REFLECT_101: edge pixel not doubled
    assert P.blur3(corner)[0, 0] == 0.25


def test_open5_ellipse_semantics():
    el = np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]], np.uint8)
    img = np.zeros((11, 11), np.uint8); img[3:8, 3:8] = el
    assert np.array_equal(P.open5(img), img)                             # the element itself survives
    img2 = img.copy(); img2[5, 5] = 0
    assert P.open5(img2).sum() == 0                                      # one missing pixel -> erased
    full = np.ones((9, 9), np.uint8)
    assert np.array_equal(P.open5(full), full)                           # border never erodes
    blob = np.zeros((12, 12), np.uint8); blob[0:5, 0:6] = 1             # touching the border
    ref = ndimage.binary_opening(np.pad(blob, 2, constant_values=1), structure=el)[2:-2, 2:-2] & (blob > 0)
    got = P.open5(blob)
    er = ndimage.binary_erosion(blob, structure=el, border_value=1)
    assert np.array_equal(got, ndimage.binary_dilation(er, structure=el))


def test_contour_and_moments_of_rectangle():
    inst = np.zeros((12, 14), np.int32); inst[2:7, 3:11] = 5           # rows 2..6, cols 3..10
    d = P.instances(inst, np.ones_like(inst, dtype=np.uint8))
    r = d[5]
    assert r["bbox"].tolist() == [[2, 3], [7, 11]]
    assert r["contour"].tolist() == [[3, 2], [3, 6], [10, 6], [10, 2]]   # (x, y): TL, BL, BR, TR (OpenCV order)
    assert np.allclose(r["centroid"], [6.5, 4.0])
    assert r["type"] == 1 and abs(r["type_prob"] - 1.0) < 1e-6
    line = np.zeros((5, 9), np.int32); line[2, 1:8] = 3
    assert 3 not in P.instances(line, np.ones_like(line, dtype=np.uint8))   # 2-point contour -> skipped (:113-116)


def test_type_vote_rules():
    inst = np.zeros((8, 8), np.int32); inst[1:7, 1:7] = 1               # 36 px
    t = np.zeros((8, 8), np.uint8)
    t[1:7, 1:7] = 0; t[1:3, 1:7] = 2; t[3:5, 1:7] = 4                   # 12 x type2, 12 x type4, 12 x type0
    r = P.instances(inst, t)[1]
    assert r["type"] == 2                                               # tie: smaller type id first; 0 wins only alone
    t2 = np.zeros((8, 8), np.uint8); t2[1:2, 1:7] = 3                   # 30 x 0, 6 x 3 -> 0 dominant -> 2nd = 3
    r = P.instances(inst, t2)[1]
    assert r["type"] == 3 and abs(r["type_prob"] - 6 / (36 + 1e-6)) < 1e-12
    r = P.instances(inst, np.zeros((8, 8), np.uint8))[1]
    assert r["type"] == 0


def test_no_background_quirk():
    inst = np.ones((6, 6), np.int32); inst[:, 3:] = 2                   # no label 0: np.unique(..)[1:] drops id 1
    d = P.instances(inst, np.ones((6, 6), np.uint8))
    assert list(d.keys()) == [2]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "postproc_*.npz"))))
def test_watershed_matches_skimage_golden(path):
    g = np.load(path)
    idx, size, k, mag = [int(v) for v in g["meta"]]
    tm, bm, hv, _ = synth_nuclei_maps(idx, size, k)
    obj, ks = (10, 21) if mag == 40 else (3, 11)
    inst, blb, dist, marker = P.proc_np_hv(bm, hv[0], hv[1], obj, ks, debug=True)
    assert np.array_equal(inst, g["skimage_watershed"])
    assert np.array_equal(inst, g["oracle_inst"])
    pm = np.stack([tm.astype(np.float32), bm.astype(np.float32), hv[0], hv[1]], -1)
    inst2, d = P.postprocess_tile(pm, 6, mag)
    assert np.array_equal(inst2, inst)
    ids = np.array(sorted(d.keys()), dtype=np.int32)
    assert np.array_equal(ids, g["ids"])
    assert np.array_equal(np.array([d[i]["bbox"].ravel() for i in ids]), g["bbox"])
    assert np.array_equal(np.array([d[i]["centroid"] for i in ids]), g["centroid"])
    assert np.array_equal(np.array([d[i]["type"] for i in ids]), g["type"])
    assert np.array_equal(np.concatenate([d[i]["contour"] for i in ids]), g["contour_cat"])


def _numpy_scipy_chain(bm, hv0, hv1, object_size, ksize):
    """Independent composition of __proc_np_hv from numpy/scipy primitives (float association order differs
    from OpenCV's filters, so continuous stages are compared with a tolerance, discrete ones by agreement)."""
    blb = (bm >= 0.5).astype(np.int32)
    lab, _ = ndimage.label(blb)
    sizes = np.bincount(lab.ravel()); lab[sizes[lab] < 10] = 0
    blb = (lab > 0).astype(np.int32)

    def nrm(a):
        return ((a.astype(np.float64) - a.min()) / (a.max() - a.min())).astype(np.float32)
    from math import comb
    sm = np.array([comb(ksize - 1, j) for j in range(ksize)], np.float64)
    dv = np.convolve([comb(ksize - 2, j) for j in range(ksize - 1)], [-1, 1]).astype(np.float64)
    h, v = nrm(hv0), nrm(hv1)
    sh = ndimage.correlate1d(ndimage.correlate1d(h.astype(np.float64), dv, axis=1, mode="mirror"), sm, axis=0, mode="mirror")
    sv = ndimage.correlate1d(ndimage.correlate1d(v.astype(np.float64), sm, axis=1, mode="mirror"), dv, axis=0, mode="mirror")
    sh, sv = 1 - nrm(sh), 1 - nrm(sv)
    overall = np.maximum(sh, sv) - (1 - blb)
    overall[overall < 0] = 0
    dist = (1.0 - overall) * blb
    k3 = np.array([0.25, 0.5, 0.25])
    dist = -ndimage.correlate1d(ndimage.correlate1d(dist, k3, axis=1, mode="mirror"), k3, axis=0, mode="mirror")
    return blb, dist, overall


def test_chain_against_numpy_scipy_composition():
    tm, bm, hv, _ = synth_nuclei_maps(5, 256, 900)
    inst, blb, dist, marker = P.proc_np_hv(bm, hv[0], hv[1], 10, 21, debug=True)
    blb2, dist2, overall2 = _numpy_scipy_chain(bm, hv[0], hv[1], 10, 21)
    assert np.array_equal(blb, blb2)
    assert np.allclose(dist, dist2, atol=1e-5)
    assert ((marker > 0) <= (blb > 0)).all()
    # every surviving instance id is a marker id and instances tile exactly the components that own a marker
    assert set(np.unique(inst)) <= set(np.unique(marker))
    assert ((inst > 0) <= (blb > 0)).all()


def test_unknown_magnification():
    with pytest.raises(NotImplementedError):
        P.postprocess_tile(np.zeros((16, 16, 4), np.float32), 6, 10)


def test_empty_and_full_tiles():
    z = np.zeros((32, 32, 4), np.float32)
    inst, d = P.postprocess_tile(z)
    assert inst.max() == 0 and d == {}
    f = np.zeros((32, 32, 4), np.float32); f[..., 1] = 1; f[..., 0] = 2
    inst, d = P.postprocess_tile(f)                                      # constant HV: normalise -> 0, one marker
    assert inst.shape == (32, 32)
