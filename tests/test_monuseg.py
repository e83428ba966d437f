"""MoNuSeg evaluation caller (reference inference_cellvit_experiment_monuseg.py:300-781): host pieces on CPU,
the three routes end to end on the GPU with a stand-in network that returns the maps of a synthetic nucleus world."""
import numpy as np
import pytest
import torch

from cellvit_amd.inference import evaluate_monuseg as EM


def test_decompose_and_reassembly_match_the_einops_formulas():
    from einops import rearrange
    g = torch.Generator().manual_seed(0)
    img = torch.rand((3, 1024, 1024), generator=g)
    a = EM.decompose(img, True, 0)
    ref = rearrange(rearrange(img, "c (h i) (w j) -> c h w i j", i=256, j=256), "c i j w h -> (i j) c w h")
    assert torch.equal(a, ref) and a.shape == (16, 3, 256, 256)
    b = EM.decompose(img, True, 64)
    ref = rearrange(img.unfold(1, 256, 192).unfold(2, 256, 192), "c i j w h -> (i j) c w h")
    assert torch.equal(b, ref) and b.shape == (25, 3, 256, 256)
    assert torch.equal(EM.decompose(img, False, 0), img[None])
    # tile (i, j) of the overlapping decomposition starts at (i * 192, j * 192) = i * 256 - i * overlap (merge_predictions, :625-626)
    assert torch.equal(b[1 * 5 + 2], img[:, 192:448, 384:640])
    pred = {k: torch.rand((16, d, 256, 256), generator=g) for k, d in (("nuclei_binary_map", 2), ("hv_map", 2), ("nuclei_type_map", 6))}
    out = EM.MoNuSegEvaluator.post_process_patching(pred)
    for k, v in pred.items():
        assert torch.equal(out[k], rearrange(v, "(i j) d w h -> d (i w) (j h)", i=4, j=4)[None])


def test_fill_poly_known_answers():
    c = np.zeros((12, 12), np.int32)
    EM.fill_poly(c, np.array([[2, 3], [2, 7], [8, 7], [8, 3]]), 5)           # axis-parallel rectangle, x 2..8, y 3..7 inclusive
    want = np.zeros((12, 12), np.int32); want[3:8, 2:9] = 5
    assert np.array_equal(c, want)
    c[:] = 0
    EM.fill_poly(c, np.array([[1, 1], [9, 1], [1, 9]]), 1)                   # right triangle incl. its hypotenuse pixels
    assert c[1, 1] == 1 and c[1, 9] == 1 and c[9, 1] == 1 and c[5, 5] == 1 and c[6, 6] == 0 and c.sum() == sum(range(1, 10))
    c[:] = 0
    EM.fill_poly(c, np.array([[-3, 4], [20, 4], [20, 6], [-3, 6]]), 2)       # clipped at the canvas
    assert (c[4:7] == 2).all() and c[:4].sum() == 0 and c[7:].sum() == 0


def _cells_of_patches(world, n=5, overlap=64):
    """Per-patch nucleus dicts as calculate_instance_map returns them, built on the host from the world's instance map
    (bbox / centroid / a rectangular contour of the visible part): input of merge_predictions without a GPU."""
    inst = world[3][:1024, :1024]
    out = []
    for i in range(n):
        for j in range(n):
            y0, x0 = i * (256 - overlap), j * (256 - overlap)
            sub = inst[y0:y0 + 256, x0:x0 + 256]
            d = {}
            for k, cid in enumerate(np.unique(sub)[1:]):
                ys, xs = np.nonzero(sub == cid)
                r0, r1, c0, c1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
                if r1 - r0 < 3 or c1 - c0 < 3:
                    continue
                d[k + 1] = {"bbox": np.array([[r0, c0], [r1, c1]]), "centroid": np.array([xs.mean(), ys.mean()]),
                            "contour": np.array([[c0, r0], [c0, r1 - 1], [c1 - 1, r1 - 1], [c1 - 1, r0]], np.int32),
                            "type_prob": 0.9, "type": 1 + int(cid) % 5}
            out.append(d)
    return out


def test_merge_predictions_equals_the_dict_restatement_of_the_reference():
    """merge_predictions (packed arrays through stitch_margin_records, host route here) keeps exactly the cells that the
    reference's own construction keeps: its per-cell dicts (:619-668) through the dict-based CellPostProcessor restatement."""
    from cellvit_amd import sharding as S
    from cellvit_amd.synth import synth_world_maps
    from oracle import stitch_ref as SR
    world = synth_world_maps(21, 1920, 2800, return_inst=True)
    patches = _cells_of_patches(world)
    ev = EM.MoNuSegEvaluator(model=None, patching=True, overlap=64, device=torch.device("cpu"))
    got = ev.merge_predictions({"nuclei_binary_map": torch.zeros((25, 2, 1, 1)), "instance_types": patches}, 64)
    cell_list = []
    for i in range(5):
        for j in range(5):
            off = np.array([i * 256 - i * 64, j * 256 - j * 64])
            for cell in patches[i * 5 + j].values():
                d = {"bbox": (cell["bbox"] + off).tolist(), "centroid": (cell["centroid"] + np.flip(off)).tolist(),
                     "contour": (cell["contour"] + np.flip(off)).tolist(), "type_prob": cell["type_prob"], "type": cell["type"],
                     "patch_coordinates": [i, j], "cell_status": S.cell_status(cell["bbox"], 256, 64), "offset_global": off.tolist()}
                if np.max(cell["bbox"]) == 256 or np.min(cell["bbox"]) == 0:
                    pos = S.cell_edge_position(cell["bbox"], 256)
                    d["edge_position"] = True
                    d["edge_information"] = {"position": pos, "edge_patches": S.edge_patches(pos, i, j)}
                else:
                    d["edge_position"] = False
                cell_list.append(d)
    keep = SR.stitch_cells(cell_list)
    assert got == [cell_list[k] for k in keep]
    assert 0 < len(got) < len(cell_list)


class _WorldNet:
    """Stand-in network: channel 0 / 1 of pixel (0, 0) of every input image carry its origin in the world (y / 4096, x / 4096);
    the outputs are confident logits of the world's maps at that crop."""
    num_nuclei_classes = 6

    def __init__(self, world, dev):
        self.tm = torch.from_numpy(world[0].astype(np.int64)).to(dev)
        self.bm = torch.from_numpy(world[1].astype(np.int64)).to(dev)
        self.hv = torch.from_numpy(world[2]).to(dev)

    def forward(self, img):
        B, _, h, w = img.shape
        outs = {"nuclei_binary_map": [], "nuclei_type_map": [], "hv_map": []}
        for b in range(B):
            y0, x0 = int(round(float(img[b, 0, 0, 0]) * 4096)), int(round(float(img[b, 1, 0, 0]) * 4096))
            sl = (slice(y0, y0 + h), slice(x0, x0 + w))
            outs["nuclei_binary_map"].append(10.0 * torch.nn.functional.one_hot(self.bm[sl], 2).permute(2, 0, 1).float())
            outs["nuclei_type_map"].append(10.0 * torch.nn.functional.one_hot(self.tm[sl], 6).permute(2, 0, 1).float())
            outs["hv_map"].append(self.hv[(slice(None),) + sl])
        return {k: torch.stack(v) for k, v in outs.items()}

    def calculate_instance_map(self, predictions, magnification=40):
        from cellvit_amd.postproc import calculate_instance_map
        return calculate_instance_map(predictions, 6, magnification)


@pytest.mark.gpu
@pytest.mark.parametrize("patching,overlap", [(False, 0), (True, 0), (True, 64)])
def test_monuseg_routes_end_to_end(patching, overlap):
    from cellvit_amd.synth import synth_world_maps
    dev = torch.device("cuda", 0)
    world = synth_world_maps(21, 1920, 2800, return_inst=True)
    inst = torch.from_numpy(world[3][:1024, :1024].astype(np.int64))
    ys, xs = torch.meshgrid(torch.arange(1024), torch.arange(1024), indexing="ij")
    img = torch.stack([ys / 4096.0, xs / 4096.0, torch.zeros((1024, 1024))]).float()
    mask = {"instance_map": inst, "nuclei_binary_map": (inst > 0).long()}
    ev = EM.MoNuSegEvaluator(_WorldNet(world, dev), 40, patching, overlap, device=dev)
    agg, per = ev.run([(img, mask, "img0.png")])
    print(f"\n[monuseg patching={patching} overlap={overlap}] {agg}")
    assert set(agg) == {"Binary-Cell-Dice-Mean", "Binary-Cell-Jacard-Mean", "bPQ", "bDQ", "bSQ", "f1_detection",
                        "precision_detection", "recall_detection"}
    # the maps ARE the ground truth (up to HV noise and the 10 % type-label noise): the metrics must say so
    assert agg["Binary-Cell-Dice-Mean"] > 0.95 and agg["bPQ"] > 0.75 and agg["f1_detection"] > 0.9, agg
