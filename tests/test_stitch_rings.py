"""Slide stitch, the reference's repair of invalid contour rings (`if not poly.is_valid: poly.buffer(0)` -> the part of the largest
area, cell_detection.py:689-704) and its tie rule (np.argmax: first of equal areas, :743-746):
  * the library's host routine `cv_stitch_repair_rings` against the oracle's lattice-chain restatement on hand-made rings
    (pinched at a corner, spur, double pinch) and on contours of random 8-connected blobs traced by the oracle's contour tracer;
  * a slide fixture in which whole-ring and largest-lobe semantics keep DIFFERENT cells: product (host route here, device route
    under -m gpu) == oracle, and != what the unrepaired ring would give;
  * first-of-equals among equal-area partners, oracle and product.
shapely itself is not installable here: GEOS' part order for equal lobes and the STRtree's query order stay unpinned."""
import numpy as np
import pytest
import torch

from cellvit_amd import sharding as S
from cellvit_amd.inference import cell_detection as CD
from cellvit_amd.inference import stitch as ST
from oracle import stitch_ref as SR

# two squares joined at the lattice point (10, 10): lobes of area 100 and 150
PINCHED = np.array([[0, 0], [10, 0], [10, 10], [20, 10], [20, 25], [10, 25], [10, 10], [0, 10]])
# a square with a one-pixel spur walked out and back
SPUR = np.array([[0, 0], [10, 0], [10, 5], [15, 5], [10, 5], [10, 10], [0, 10]])
# three lobes on a diagonal chain, the middle one largest
CHAIN3 = np.array([[0, 0], [4, 0], [4, 4], [12, 4], [12, 12], [15, 12], [15, 15], [12, 15], [12, 12], [4, 12], [4, 4], [0, 4]])
SIMPLE = np.array([[0, 0], [6, 0], [6, 2], [2, 2], [2, 6], [0, 6]])


def _repair(rings):
    off = np.zeros(len(rings) + 1, np.int64)
    np.cumsum([len(r) for r in rings], out=off[1:])
    ct = np.concatenate(rings).astype(np.int32)
    o2, c2, nrep = ST.repair_rings(off, ct)
    return [c2[o2[i]:o2[i + 1]] for i in range(len(rings))], nrep


def _chain_set(ring):
    return set(SR._lattice_chain(ring))


def test_repair_rings_known_answers():
    out, nrep = _repair([PINCHED, SPUR, CHAIN3, SIMPLE])
    assert nrep == 3
    assert SR._poly_area(out[0]) == 150.0 and _chain_set(out[0]) == _chain_set(np.array([[10, 10], [20, 10], [20, 25], [10, 25]]))
    assert SR._poly_area(out[1]) == 100.0 and len(out[1]) == 4            # the spur is gone, collinear points removed
    assert SR._poly_area(out[2]) == 64.0
    assert np.array_equal(out[3], SIMPLE)                                  # simple rings pass through untouched
    for ring, got in zip([PINCHED, SPUR, CHAIN3, SIMPLE], out):
        want = SR.largest_lobe(ring)
        assert SR._poly_area(got) == SR._poly_area(want) and _chain_set(got) == _chain_set(want)
        assert SR.ring_is_simple(got)
        assert len(got) <= len(ring)
    assert not SR.ring_is_simple(PINCHED) and not SR.ring_is_simple(SPUR) and SR.ring_is_simple(SIMPLE)


def test_repair_rings_on_traced_contours_of_random_blobs():
    """Contours of random 8-connected blobs (diagonal pinches and spurs included) from the oracle's contour tracer."""
    from scipy import ndimage
    from oracle import postproc_ref as PR
    rng = np.random.default_rng(7)
    rings = []
    for k in range(60):
        img = np.zeros((40, 40), np.int32)
        for _ in range(int(rng.integers(2, 6))):             # a few small rectangles touching at corners / through 1-px bridges
            y, x = rng.integers(4, 30, 2)
            h, w = rng.integers(1, 8, 2)
            img[y:y + h, x:x + w] = 1
        y, x = rng.integers(6, 30, 2)
        for d in range(int(rng.integers(2, 9))):             # a diagonal staircase: 8-connected, pinched at every step
            img[y + d, x + d] = 1
        lab, _ = ndimage.label(img, structure=np.ones((3, 3), int))       # one instance id per 8-connected blob
        recs = PR.instances(lab, np.ones((40, 40), np.int32), 2)
        for c in recs.values():
            if len(c["contour"]) >= 3:
                rings.append(np.asarray(c["contour"], np.int64))
    assert len(rings) >= 30
    out, nrep = _repair(rings)
    n_bad = 0
    for ring, got in zip(rings, out):
        want = SR.largest_lobe(ring)
        if not SR.ring_is_simple(ring):
            n_bad += 1
            assert SR.ring_is_simple(got)
        assert SR._poly_area(got) == SR._poly_area(want)
        assert _chain_set(got) == _chain_set(want)
    assert n_bad == nrep and n_bad >= 10, (n_bad, nrep)


def _cell(contour, row, col, status=4):
    contour = np.asarray(contour)
    r0, c0, r1, c1 = contour[:, 1].min(), contour[:, 0].min(), contour[:, 1].max() + 1, contour[:, 0].max() + 1
    return {"bbox": np.array([[r0, c0], [r1, c1]]), "contour": contour, "row": row, "col": col, "status": status}


def _records(cells):
    irs, frs, cts = [], [], []
    for k, c in enumerate(cells):
        bb = c["bbox"]
        irs.append([c["row"], c["col"], bb[0, 0], bb[0, 1], bb[1, 0], bb[1, 1], 1, c["status"], 0, c["row"] * 3 + c["col"], len(c["contour"]), k + 1])
        frs.append([0.0, 0.0, 1.0])
        cts.append(c["contour"])
    return CD.SlideCells(np.asarray(irs, np.int32), np.asarray(frs, np.float64), np.concatenate(cts).astype(np.int32),
                         torch.zeros((len(irs), 4)))


def _fixture():
    """Tile (0,0), local coordinates.  Cell 0: PINCHED shifted to (500, 900) — lobes of 100 px (upper left) and 150 px.  Cell 1: a
    4 x 4 square inside the SMALL lobe: it overlaps the whole ring (16 / 16 of its own area) but not the ring's largest lobe —
    with the reference's repair the two cells do not overlap and both survive; on the unrepaired ring (even-odd area of both
    lobes) they overlap by 100 % of cell 1.  Cell 2: an unrelated margin cell."""
    pin = PINCHED + np.array([900, 500])
    small = np.array([[3, 3], [7, 3], [7, 7], [3, 7]]) + np.array([900, 500])
    far = np.array([[0, 0], [30, 0], [30, 30], [0, 30]]) + np.array([300, 950])          # an unrelated margin cell
    return [_cell(pin, 0, 0), _cell(small, 0, 0), _cell(far, 0, 0)]


def _keep_product(sc, device=None):
    return ST.stitch_margin_records(sc.ir, sc.ct, 1024, 1, 64, device=device).tolist()


def _dicts(sc):
    from test_cli import _to_dicts_scalar
    return _to_dicts_scalar(sc, 1024, 1, 64)


def test_pinched_ring_changes_the_stitch_and_product_equals_oracle():
    sc = _records(_fixture())
    want = SR.stitch_cells(_dicts(sc))
    assert want == [0, 1, 2]                                # repaired: the small square lies outside the largest lobe
    assert _keep_product(sc) == want
    # the same cells WITHOUT the repair (even-odd area of the whole ring): cell 1 overlaps cell 0 and replaces it
    bbox, off, ctg = ST.global_geometry(sc.ir, sc.ct, 1024, 1, 64)
    assert ST.intersection_area(ctg[off[0]:off[1]], ctg[off[1]:off[2]]) == 16.0
    o2, c2, nrep = ST.repair_rings(off, ctg)
    assert nrep == 1 and ST.intersection_area(c2[o2[0]:o2[1]], c2[o2[1]:o2[2]]) == 0.0
    assert ST.poly_area(c2[o2[0]:o2[1]]) == 150.0


def test_first_of_equal_areas_survives():
    """Cell 0 overlaps two partners of EQUAL area: np.argmax keeps the first (cell_detection.py:743-746)."""
    base = np.array([[0, 0], [20, 0], [20, 20], [0, 20]]) + np.array([900, 500])
    a = np.array([[0, 0], [10, 0], [10, 10], [0, 10]]) + np.array([895, 495])           # over one corner of cell 0
    b = np.array([[0, 0], [10, 0], [10, 10], [0, 10]]) + np.array([915, 515])           # over the opposite corner; a and b disjoint
    sc = _records([_cell(base, 0, 0), _cell(a, 0, 0), _cell(b, 0, 0)])
    want = SR.stitch_cells(_dicts(sc))
    assert want == [1]                                      # partners 1 and 2 tie: the first survives, cell 0 and 2 go
    assert _keep_product(sc) == want


@pytest.mark.gpu
def test_device_ring_flags_and_stitch_on_the_fixture():
    dev = torch.device("cuda", 0)
    sc = _records(_fixture())
    assert _keep_product(sc, dev) == SR.stitch_cells(_dicts(sc)) == [0, 1, 2]
    # the flag kernel on a mix of simple and invalid rings
    rings = [PINCHED, SIMPLE, SPUR, CHAIN3, SIMPLE + 7, np.array([[0, 0], [9, 0], [9, 9], [0, 9]])]
    off = np.zeros(len(rings) + 1, np.int64)
    np.cumsum([len(r) for r in rings], out=off[1:])
    ct = np.concatenate(rings).astype(np.int32)
    import ctypes as C
    from cellvit_amd import _lib
    lib = _lib.load()
    d_off, d_ct = torch.from_numpy(off).to(dev), torch.from_numpy(ct).to(dev)
    d_fl = torch.empty(len(rings), dtype=torch.uint8, device=dev)
    _lib.check(lib.cv_stitch_ring_flags(d_off.data_ptr(), d_ct.data_ptr(), len(rings), d_fl.data_ptr(),
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    assert d_fl.cpu().tolist() == [1, 0, 1, 1, 0, 0]
    assert [not SR.ring_is_simple(r) for r in rings] == [True, False, True, True, False, False]
    # capacity overflow of the pair list is retried with the exact count, not raised
    bbox, off2, ctg = ST.global_geometry(sc.ir, sc.ct, 1024, 1, 64)
    p1, i1, a1 = ST.overlaps_device(bbox, off2, ctg, dev, cap=1)
    p2, i2, a2 = ST.overlaps_device(bbox, off2, ctg, dev)
    assert np.array_equal(p1, p2) and np.array_equal(i1, i2) and len(p1) >= 1
