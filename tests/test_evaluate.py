"""Evaluation caller: the metric restatements against outputs of the imported reference
(tests/golden/eval_cases.npz, tools/make_golden_eval.py), the centroid pairing against an exhaustive search, the step /
dataset aggregation on hand-built cases (CPU), and — on the GPU — `calculate_instances` (cv_pp_records) against the
oracle's per-instance records plus the whole caller around the C-ABI post-processing."""
import itertools
import os

import numpy as np
import pytest
import torch

from cellvit_amd.metrics import (binarize, binary_dice, binary_jaccard, cell_detection_scores, cell_type_detection_scores,
                                 pair_coordinates)
from cellvit_amd.synth import synth_nuclei_maps

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_cases.npz"))
CFG = {"tissue_types": {"Breast": 0, "Colon": 1}, "nuclei_types": {"Background": 0, "Neoplastic": 1, "Inflammatory": 2,
                                                                     "Connective": 3, "Dead": 4, "Epithelial": 5}}


@pytest.mark.parametrize("c", range(int(G["n_bin"])))
def test_binarize_matches_reference(c):
    out = binarize(G[f"bin{c}_x"])
    assert out.dtype == np.int32 and np.array_equal(out, G[f"bin{c}_out"])


@pytest.mark.parametrize("c", range(int(G["n_det"])))
def test_detection_scores_match_reference(c):
    pt, pp, ut, up = (G[f"det{c}_{k}"] for k in ("pt", "pp", "ut", "up"))
    assert np.allclose(cell_detection_scores(pt, pp, ut, up), G[f"det{c}_all"], rtol=0, atol=1e-15)
    for t in range(1, 6):
        assert np.allclose(cell_type_detection_scores(pt, pp, ut, up, t), G[f"det{c}_type"][t - 1], rtol=0, atol=1e-15)
        assert np.allclose(cell_type_detection_scores(pt, pp, ut, up, t, exhaustive=False), G[f"det{c}_type_nonex"][t - 1],
                           rtol=0, atol=1e-15)


def test_pair_coordinates_is_the_minimum_cost_unique_pairing():
    rng = np.random.default_rng(5)
    for trial in range(20):
        na, nb = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        a, b = rng.uniform(0, 40, (na, 2)), rng.uniform(0, 40, (nb, 2))
        pairing, ua, ub = pair_coordinates(a, b, 12)
        d = np.linalg.norm(a[:, None] - b[None], axis=-1)
        # exhaustive minimum total distance over all injective maps of the smaller set into the larger
        if na <= nb:
            best = min(itertools.permutations(range(nb), na), key=lambda p: sum(d[i, p[i]] for i in range(na)))
            full = [(i, best[i]) for i in range(na)]
        else:
            best = min(itertools.permutations(range(na), nb), key=lambda p: sum(d[p[j], j] for j in range(nb)))
            full = [(best[j], j) for j in range(nb)]
        want = sorted((i, j) for i, j in full if d[i, j] <= 12)
        assert sorted(map(tuple, pairing.tolist())) == want
        assert sorted(ua.tolist()) == sorted(set(range(na)) - {i for i, _ in want})
        assert sorted(ub.tolist()) == sorted(set(range(nb)) - {j for _, j in want})


def test_binary_dice_and_jaccard():
    a = np.zeros((6, 6), np.int64); a[1:4, 1:4] = 1
    b = np.zeros((6, 6), np.int64); b[2:5, 2:5] = 1
    assert binary_dice(a, b) == pytest.approx(2 * 4 / 18) and binary_jaccard(a, b) == pytest.approx(4 / 14)
    assert binary_dice(a, a) == 1.0 and binary_jaccard(a, a) == 1.0
    assert binary_dice(np.zeros_like(a), np.zeros_like(a)) == 0.0 and binary_jaccard(np.zeros_like(a), np.zeros_like(a)) == 0.0


def _fake_dicts(inst, types):
    """host-side stand-in for the record dicts (centroid + majority type) so the step metrics can be tested without a GPU"""
    d = {}
    for i in np.unique(inst)[1:]:
        ys, xs = np.nonzero(inst == i)
        vals, cnt = np.unique(types[ys, xs], return_counts=True)
        d[int(i)] = {"centroid": np.array([xs.mean(), ys.mean()]), "type": int(vals[np.argmax(cnt)])}
    return d


def _case(seed, size=128, drop=0):
    types, binary, hv, inst = synth_nuclei_maps(seed, size, 30 * 64)
    ids = np.unique(inst)[1:]
    cls = {int(i): 1 + int(i) % 5 for i in ids}                  # one class per instance
    tmap = np.zeros_like(inst)
    for i, c in cls.items():
        tmap[inst == i] = c
    pred_inst = inst.copy()
    for i in ids[:drop]:
        pred_inst[pred_inst == i] = 0
    return inst, tmap, pred_inst, hv


def _stack(cases, num_classes=6):
    """(predictions, gt) dicts in the layout unpack_predictions / unpack_masks produce, on the CPU"""
    B = len(cases)
    H, W = cases[0][0].shape
    pred = {"tissue_types": torch.zeros(B, 2), "nuclei_binary_map": torch.zeros(B, 2, H, W), "instance_types": [],
            "instance_types_nuclei": torch.zeros(B, num_classes, H, W), "instance_map": torch.zeros(B, H, W)}
    gt = {"tissue_types": torch.zeros(B, dtype=torch.long), "nuclei_binary_map": torch.zeros(B, 2, H, W), "instance_types": [],
          "instance_types_nuclei": torch.zeros(B, num_classes, H, W), "instance_map": torch.zeros(B, H, W, dtype=torch.int64)}
    for b, (inst, tmap, pinst, _) in enumerate(cases):
        pred["tissue_types"][b, b % 2] = 1.0
        gt["tissue_types"][b] = 0
        for d, im in ((pred, pinst), (gt, inst)):
            d["nuclei_binary_map"][b, 1] = torch.from_numpy((im > 0).astype(np.float32))
            d["nuclei_binary_map"][b, 0] = 1 - d["nuclei_binary_map"][b, 1]
            d["instance_map"][b] = torch.from_numpy(im)
            for c in range(num_classes):
                d["instance_types_nuclei"][b, c] = torch.from_numpy(np.where(tmap == c, im, 0).astype(np.float32))
            d["instance_types"].append(_fake_dicts(im, tmap))
    return pred, gt


def test_step_metrics_and_aggregation_on_hand_built_batches():
    from cellvit_amd.inference.evaluate import PatchEvaluator
    ev = PatchEvaluator(model=None, dataset_config=CFG, magnification=40, device=torch.device("cpu"))
    perfect = [_case(10), _case(11)]
    pred, gt = _stack(perfect)
    bm, scores = ev.calculate_step_metric(pred, gt, ["a", "b"])
    assert bm["binary_dice_scores"] == [1.0, 1.0] and bm["binary_jaccard_scores"] == [1.0, 1.0]
    assert np.allclose(bm["pq_scores"], 1.0, atol=1e-5) and np.allclose(bm["dq_scores"], 1.0, atol=1e-5)
    assert np.isnan(bm["cell_type_pq_scores"][0][0])                    # background "class": never in the ground truth
    assert np.allclose(bm["cell_type_pq_scores"][0][1:], 1.0, atol=1e-5)
    n0, n1 = (len(d) for d in gt["instance_types"])
    assert bm["paired_all"].shape == (n0 + n1, 2) and bm["unpaired_true_all"].size == 0 and bm["unpaired_pred_all"].size == 0
    assert np.array_equal(bm["paired_all"][n0:, 0], np.arange(n0, n0 + n1))     # second image offset by the first's counts
    assert list(bm["tissue_pred"]) == [0, 1] and list(bm["tissue_gt"]) == [0, 0]

    # three instances missed in the prediction of one image: recall drops, precision stays 1, bPQ = tp / (tp + fn / 2)
    pred2, gt2 = _stack([_case(12, drop=3)])
    bm2, _ = ev.calculate_step_metric(pred2, gt2, ["c"])
    n = len(gt2["instance_types"][0])
    assert bm2["unpaired_true_all"].size == 3 and bm2["unpaired_pred_all"].size == 0
    assert bm2["dq_scores"][0] == pytest.approx((n - 3) / ((n - 3) + 1.5), abs=1e-5)
    assert bm2["binary_dice_scores"][0] < 1.0

    acc = {k: list(bm[k]) + list(bm2[k]) for k in ("image_names", "binary_dice_scores", "binary_jaccard_scores", "pq_scores", "dq_scores",
                                                   "sq_scores", "cell_type_pq_scores", "cell_type_dq_scores", "cell_type_sq_scores")}
    acc["tissue_types"] = ["Breast", "Colon", "Breast"]
    t_off, p_off = bm["true_inst_type_all"].shape[0], bm["pred_inst_type_all"].shape[0]
    res = ev.aggregate(acc, [bm["tissue_pred"], bm2["tissue_pred"]], [bm["tissue_gt"], bm2["tissue_gt"]],
                       [bm["paired_all"], bm2["paired_all"] + np.array([[t_off, p_off]])],
                       [bm["unpaired_true_all"], bm2["unpaired_true_all"] + t_off],
                       [bm["unpaired_pred_all"], bm2["unpaired_pred_all"] + p_off],
                       [bm["true_inst_type_all"], bm2["true_inst_type_all"]], [bm["pred_inst_type_all"], bm2["pred_inst_type_all"]])
    ds = res["dataset"]
    tot = n0 + n1 + n
    assert ds["precision_detection"] == 1.0 and ds["recall_detection"] == pytest.approx((tot - 3) / tot)
    assert ds["f1_detection"] == pytest.approx(2 * (tot - 3) / (2 * (tot - 3) + 3))
    assert ds["Tissue-Multiclass-Accuracy"] == pytest.approx(2 / 3)
    assert ds["bPQ"] == pytest.approx(np.mean(acc["pq_scores"]))
    assert set(res["tissue_metrics"]) == {"breast", "colon"} and res["tissue_metrics"]["colon"]["bPQ"] == pytest.approx(1.0, abs=1e-5)
    assert set(res["nuclei_metrics_pq"]) == {"Neoplastic", "Inflammatory", "Connective", "Dead", "Epithelial"}
    assert all(0.0 < res["nuclei_metrics_d"][k]["f1_cell"] <= 1.0 for k in res["nuclei_metrics_d"])
    assert set(res["image_metrics"]) == {"a", "b", "c"}


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_calculate_instances_equals_the_oracle_records():
    """cv_pp_records on given instance maps == the oracle's P7/P8 restatement, dict for dict (bit-exact)."""
    from oracle import postproc_ref as O
    from cellvit_amd.postproc import calculate_instances
    insts, tmaps = [], []
    for s in range(3):
        inst, tmap, _, _ = _case(40 + s, size=256)
        if s == 1:
            inst = inst * 3                                       # non-contiguous ids
            inst[5, 5:9] = 2000                                   # a 4-pixel line: contour of 2 points -> dropped from the dict
        if s == 2:
            inst[inst == 0] = -1                                  # no zero in the map: np.unique()[1:] drops the smallest id
            inst = np.where(inst < 0, int(inst.max()) + 5, inst)  # ... so paint the "background" with the LARGEST id instead
        insts.append(inst); tmaps.append(tmap)
    inst_t = torch.from_numpy(np.stack(insts)).cuda()
    onehot = torch.nn.functional.one_hot(torch.from_numpy(np.stack(tmaps)).long(), 6).permute(0, 3, 1, 2).float().cuda()
    got = calculate_instances(onehot, inst_t)
    for b in range(3):
        want = O.instances(insts[b], tmaps[b], 6)
        assert list(got[b].keys()) == list(want.keys()) and len(want) > 10
        for k in want:
            assert np.array_equal(got[b][k]["bbox"], want[k]["bbox"])
            assert np.array_equal(got[b][k]["centroid"], want[k]["centroid"])
            assert np.array_equal(got[b][k]["contour"], want[k]["contour"])
            assert got[b][k]["type"] == want[k]["type"] and got[b][k]["type_prob"] == want[k]["type_prob"]
    assert 2000 not in got[1] and 2000 in np.unique(insts[1])
    assert int(np.unique(insts[2])[0]) not in got[2]
    assert torch.equal(inst_t, torch.from_numpy(np.stack(insts)).cuda())          # the caller's map is not modified


@pytest.mark.gpu
def test_calculate_instances_with_more_than_eight_classes():
    """post_proc_cellvit.py:300-318 votes with np.unique (no class limit): 12 and 19 classes on the device (windows of 8 types)
    == the oracle, incl. equal counts (the lower type wins) and background replaced by a runner-up above type 7; the
    prediction path (`cv_pp_run`) takes the same vote."""
    from oracle import postproc_ref as O
    from cellvit_amd.postproc import calculate_instances, postprocess_device, records_to_dicts, _params
    rng = np.random.default_rng(5)
    for nr in (12, 19):
        insts, tmaps = [], []
        for s in range(2):
            inst, tmap, _, _ = _case(60 + s, size=256)
            tm = rng.integers(0, nr, tmap.shape).astype(np.uint8)            # noisy votes over all classes
            ids = np.unique(inst)[1:]
            for k, i in enumerate(ids):                                      # most instances: one dominant class, many above 7
                m = inst == i
                sel = m & (rng.random(inst.shape) < 0.6)
                tm[sel] = (k * 5 + 3) % nr
            if len(ids) > 3:
                ys, xs = np.nonzero(inst == ids[0])                          # exact tie between types 9 and 11: the lower wins
                tm[ys, xs] = np.where(np.arange(len(ys)) % 2 == 0, 9, 11)[:len(ys)] if len(ys) % 2 == 0 else tm[ys, xs]
                ys, xs = np.nonzero(inst == ids[1])                          # background majority, runner-up type 10
                tm[ys, xs] = 0
                tm[ys[: max(1, len(ys) // 3)], xs[: max(1, len(ys) // 3)]] = 10
            insts.append(inst); tmaps.append(tm)
        inst_t = torch.from_numpy(np.stack(insts)).cuda()
        onehot = torch.nn.functional.one_hot(torch.from_numpy(np.stack(tmaps)).long(), nr).permute(0, 3, 1, 2).float().cuda()
        got = calculate_instances(onehot, inst_t)
        seen = set()
        for b in range(2):
            want = O.instances(insts[b], tmaps[b], nr)
            assert list(got[b].keys()) == list(want.keys()) and len(want) > 10
            for k in want:
                assert got[b][k]["type"] == want[k]["type"] and got[b][k]["type_prob"] == want[k]["type_prob"], (nr, b, k)
                assert np.array_equal(got[b][k]["contour"], want[k]["contour"])
                seen.add(want[k]["type"])
        assert max(seen) > 8 and len(seen) > 6
    # prediction path with 12 classes: same records as the oracle's full chain
    nr = 12
    inst, tmap, bm, hvm = _case(70, size=256)
    tm = ((tmap.astype(np.int64) * 5) % nr).astype(np.uint8)
    obj, ks = _params(40)
    dev = torch.device("cuda", 0)
    i_d, recs, n_recs, contours, n_pts = postprocess_device(torch.from_numpy((inst > 0).astype(np.uint8))[None].to(dev), torch.from_numpy(tm)[None].to(dev),
                                                            torch.from_numpy(np.ascontiguousarray(hvm))[None].to(dev), nr, obj, ks)
    got = records_to_dicts(recs, n_recs, contours, n_pts)[0]
    pm = np.stack([tm.astype(np.float32), (inst > 0).astype(np.float32), hvm[0], hvm[1]], -1)
    i_o, want = O.postprocess_tile(pm, nr, 40)
    assert np.array_equal(i_d[0].cpu().numpy(), i_o) and list(got.keys()) == list(want.keys()) and len(want) > 5
    for k in want:
        assert got[k]["type"] == want[k]["type"] and got[k]["type_prob"] == want[k]["type_prob"]


class _MapsModel:
    """Stand-in network: returns logits whose argmax / HV maps are the synthetic ground truth, then the REAL
    calculate_instance_map / generate_instance_nuclei_map of the shim (C-ABI post-processing)."""

    def __init__(self, cases):
        from cellvit_amd.model import CellViT256
        self.m = CellViT256(None, 6, 2)
        self.cases = cases
        self.calculate_instance_map = self.m.calculate_instance_map
        self.generate_instance_nuclei_map = self.m.generate_instance_nuclei_map

    def forward(self, imgs):
        B = imgs.shape[0]
        H, W = self.cases[0][0].shape
        nb, nt, hv = torch.zeros(B, 2, H, W), torch.zeros(B, 6, H, W), torch.zeros(B, 2, H, W)
        for b, (inst, tmap, _, hvm) in enumerate(self.cases[:B]):
            fg = torch.from_numpy((inst > 0).astype(np.float32))
            nb[b, 1], nb[b, 0] = 4 * fg, 4 * (1 - fg)
            nt[b] = 4 * torch.nn.functional.one_hot(torch.from_numpy(tmap).long(), 6).permute(2, 0, 1).float()
            hv[b] = torch.from_numpy(hvm)
        tt = torch.zeros(B, 2); tt[:, 1] = 1
        return {"tissue_types": tt.cuda(), "nuclei_binary_map": nb.cuda(), "hv_map": hv.cuda(), "nuclei_type_map": nt.cuda()}


@pytest.mark.gpu
def test_patch_evaluator_end_to_end_on_synthetic_ground_truth(tmp_path):
    from cellvit_amd.inference.evaluate import PatchEvaluator
    cases = []
    for s in (50, 51, 52, 53):
        types, binary, hv, inst = synth_nuclei_maps(s, 256, 30 * 16)
        ids = np.unique(inst)[1:]
        tmap = np.zeros_like(inst)
        for i in ids:
            tmap[inst == i] = 1 + int(i) % 5
        cases.append((inst, tmap, inst, hv))
    ev = PatchEvaluator(_MapsModel(cases), CFG, magnification=40)
    masks = {"nuclei_binary_map": torch.from_numpy(np.stack([(c[0] > 0) for c in cases])).long(),
             "nuclei_type_map": torch.from_numpy(np.stack([c[1] for c in cases])).long(),
             "instance_map": torch.from_numpy(np.stack([c[0] for c in cases])).long(),
             "hv_map": torch.from_numpy(np.stack([c[3] for c in cases]))}
    batch = (torch.zeros(4, 3, 256, 256), masks, ["Breast", "Colon", "Colon", "Breast"], ["p0", "p1", "p2", "p3"])
    res = ev.run([batch], outdir=tmp_path)
    ds = res["dataset"]
    # the watershed re-separates touching synthetic nuclei almost perfectly from the noisy HV maps
    assert ds["Binary-Cell-Dice-Mean"] > 0.99 and ds["bPQ"] > 0.8 and ds["mPQ"] > 0.7
    assert ds["f1_detection"] > 0.9 and ds["Tissue-Multiclass-Accuracy"] == 0.5
    assert (tmp_path / "inference_results.json").exists()
    assert set(res["image_metrics"]) == {"p0", "p1", "p2", "p3"}
