"""GPU: the slide-level de-duplication kernels (cv_stitch_overlaps: bbox-grid candidate pairs, exact polygon areas, exact
polygon-intersection areas) against the host routines of the same module and against the dict-based restatement of the
reference's CellPostProcessor (oracle/stitch_ref.py), on (a) the synthetic 3 x 3 slide of rectangles, (b) REAL contours:
the tiles of a periodic synthetic nucleus world through the device post-processing, so that neighbouring tiles see the
same nuclei in their 64-px overlap; (c) a slide-sized record set (timing)."""
import time

import numpy as np
import pytest
import torch

from cellvit_amd import sharding as S
from cellvit_amd.inference import cell_detection as CD
from cellvit_amd.inference import stitch as ST
from oracle import stitch_ref as SR

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _check_device_against_host(margin):
    bbox, off, ctg = ST.global_geometry(margin.ir, margin.ct, 1024, 1, 64)
    p_h, i_h, a_h = ST.overlaps_host(bbox, off, ctg)
    p_d, i_d, a_d = ST.overlaps_device(bbox, off, ctg, DEV)
    assert np.array_equal(p_d, p_h)                          # the same candidate pairs, each exactly once
    assert np.array_equal(a_d, a_h)                          # polygon areas: integer shoelace, exact
    assert (i_d >= 0).all()
    assert np.abs(i_d - i_h).max() <= 1e-9 * max(1.0, float(a_h.max()))
    alive = ST.edge_rule(margin.ir, 1024)
    k_d = ST.select_rounds(p_d, i_d, a_d, alive)
    k_h = ST.select_rounds(p_h, i_h, a_h, alive)
    assert np.array_equal(k_d, k_h)
    return p_d, i_d, a_d


def test_device_stitch_on_the_synthetic_rectangle_slide():
    from test_cli import _slide_cells_of, _synthetic_slide_tiles, _to_dicts_scalar
    tiles = _synthetic_slide_tiles(seed=1, n_cells=1500)
    sc = _slide_cells_of(tiles, list(range(9)))
    is_margin = sc.ir[:, S.I_STATUS] != 0
    m_idx = np.nonzero(is_margin)[0]
    margin = sc.select(m_idx)
    _check_device_against_host(margin)
    keep_m = ST.stitch_margin_records(margin.ir, margin.ct, 1024, 1, 64, device=DEV)
    keep = sorted(np.nonzero(~is_margin)[0].tolist() + m_idx[keep_m].tolist())
    assert keep == SR.stitch_cells(_to_dicts_scalar(sc, 1024, 1, 64))
    allc, dicts = CD.finalize_slide(sc, 1024, 1, 64, device=DEV)          # the product entry, device route
    assert len(dicts) == len(keep)


def _world_slide(grid=3, seed=11):
    """Records of a grid x grid slide cut from a periodic nucleus world, through the device post-processing."""
    from cellvit_amd.postproc import _params, postprocess_device
    from cellvit_amd._lib import REC_DTYPE
    from cellvit_amd.synth import synth_world_maps, world_tile
    world = synth_world_maps(seed, 1920, 2800)
    obj, ks = _params(40)
    parts = []
    for t in range(grid * grid):
        row, col = divmod(t, grid)
        tm, bm, hv = world_tile(world, row, col)
        inst, recs, n_recs, contours, n_pts = postprocess_device(torch.from_numpy(bm)[None].to(DEV), torch.from_numpy(tm)[None].to(DEV),
                                                                 torch.from_numpy(hv)[None].to(DEV), 6, obj, ks)
        torch.cuda.synchronize()
        nr = int(n_recs[0])
        rec_h = recs[0, :nr].cpu().numpy().view(REC_DTYPE).reshape(nr)
        ir, fr, ct, keep = CD.SlideCells.from_tile_records(rec_h, contours[0].cpu().numpy(), t, row, col, 0)
        parts.append(CD.SlideCells(ir, fr, ct, torch.zeros((len(ir), 4))))
    return CD.SlideCells.concat(parts)


def test_device_stitch_on_real_contours_of_overlapping_tiles():
    from test_cli import _to_dicts_scalar
    sc = _world_slide()
    is_margin = sc.ir[:, S.I_STATUS] != 0
    m_idx = np.nonzero(is_margin)[0]
    margin = sc.select(m_idx)
    pairs, inter, area = _check_device_against_host(margin)
    # neighbouring tiles really do see the same nuclei: many pairs overlap almost completely
    frac = inter / np.minimum(area[pairs[:, 0]], area[pairs[:, 1]])
    assert (frac > 0.9).sum() > 100, (len(pairs), int((frac > 0.9).sum()))
    keep_m = ST.stitch_margin_records(margin.ir, margin.ct, 1024, 1, 64, device=DEV)
    keep = sorted(np.nonzero(~is_margin)[0].tolist() + m_idx[keep_m].tolist())
    want = SR.stitch_cells(_to_dicts_scalar(sc, 1024, 1, 64))
    assert keep == want
    print(f"\n[stitch] 3x3 world slide: {len(sc)} cells, {len(m_idx)} margin, {len(pairs)} candidate pairs, "
          f"{int((frac > 0.01).sum())} overlapping; {len(keep)} kept == dict oracle")


def test_device_stitch_at_slide_scale():
    """~3e5 margin records (SURVEY §8e's estimate for a gigapixel slide): the world slide's margin set replicated on a grid
    of super-tiles.  Checks the pair list against the vectorised host pair finder and reports the time of the whole stitch."""
    sc = _world_slide()
    margin = sc.select(np.nonzero(sc.ir[:, S.I_STATUS] != 0)[0])
    reps = max(1, int(np.ceil(300000 / max(1, len(margin)))))
    side = int(np.ceil(np.sqrt(reps)))
    irs, cts = [], []
    for r in range(reps):
        ir = margin.ir.copy()
        ir[:, S.I_ROW] += 3 * (r // side) * 2            # super-tiles 2 tile rows apart: no cross-talk between replicas
        ir[:, S.I_COL] += 3 * (r % side) * 2
        ir[:, S.I_TILE] += 9 * r
        irs.append(ir); cts.append(margin.ct)
    ir, ct = np.concatenate(irs), np.concatenate(cts)
    order = S.canonical_order(ir)
    ir, _, ct = S.reorder_records(ir, np.zeros((len(ir), 3)), ct, order)
    torch.cuda.synchronize()
    ST.stitch_margin_records(ir[:1000], ct[:int(ir[:1000, S.I_CLEN].sum())], 1024, 1, 64, device=DEV)   # warm-up (library load, allocator)
    t0 = time.perf_counter()
    keep = ST.stitch_margin_records(ir, ct, 1024, 1, 64, device=DEV)
    dt = time.perf_counter() - t0
    bbox, off, ctg = ST.global_geometry(ir, ct, 1024, 1, 64)
    p_d, i_d, a_d = ST.overlaps_device(bbox, off, ctg, DEV)
    assert np.array_equal(p_d, ST.candidate_pairs_host(bbox))
    # every replica keeps the same cells
    per = len(margin)
    k0 = keep[keep < per]
    assert len(keep) == reps * len(k0)
    print(f"\n[stitch] {len(ir)} margin records, {len(p_d)} candidate pairs: device stitch {dt:.3f} s "
          f"({len(ir) / dt / 1e3:.0f} k records/s), {len(keep)} kept")
    assert dt < 2.0
