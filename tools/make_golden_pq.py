"""Generate tests/golden/pq_cases.npz: inputs + outputs of the REFERENCE panoptic-quality functions
(`cell_segmentation/utils/metrics.py`, imported here in the dev container; numpy + scipy only) on seeded random
instance maps.  The reference never travels: only this data file does."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cellvit_amd.synth import synth_nuclei_maps  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_metrics", "/root/reference/cell_segmentation/utils/metrics.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def perturb(inst, rng):
    """a 'prediction': shift, drop a few instances, merge two, add a spurious blob"""
    out = np.roll(inst, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), axis=(0, 1)).copy()
    ids = [i for i in np.unique(out) if i]
    for i in rng.choice(ids, size=min(3, len(ids)), replace=False):
        out[out == i] = 0
    ids = [i for i in np.unique(out) if i]
    if len(ids) > 2:
        out[out == ids[1]] = ids[0]
    y, x = rng.integers(0, out.shape[0] - 12, 2)
    out[y:y + 10, x:x + 9][out[y:y + 10, x:x + 9] == 0] = out.max() + 7
    return out


def main():
    rng = np.random.default_rng(2024)
    data = {}
    n = 0
    for case in range(6):
        _, _, _, gt = synth_nuclei_maps(100 + case, 256, 40 + 10 * case)
        pred = perturb(gt, rng)
        if case == 4:
            pred = np.zeros_like(gt)            # nothing predicted
        if case == 5:
            pred = np.where(pred > 0, pred, 999)  # no background in the prediction (handled by the reference, :71-72)
        for by_size in (False, True):
            t, p = ref.remap_label(gt, by_size=by_size), ref.remap_label(pred, by_size=by_size)
            for thr in (0.5, 0.3):
                (dq, sq, pq), pairs = ref.get_fast_pq(t, p, match_iou=thr)
                k = f"c{n}"
                data[k + "_true"] = gt.astype(np.int32); data[k + "_pred"] = pred.astype(np.int32)
                data[k + "_bysize"] = np.array(by_size); data[k + "_thr"] = np.array(thr)
                data[k + "_true_remap"] = t.astype(np.int32); data[k + "_pred_remap"] = p.astype(np.int32)
                data[k + "_stats"] = np.array([dq, sq, pq], dtype=np.float64)
                for nm, arr in zip(("pt", "pp", "ut", "up"), pairs):
                    data[k + "_" + nm] = np.asarray(list(arr), dtype=np.int64)
                n += 1
    data["n"] = np.array(n)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pq_cases.npz"), **data)
    print("cases:", n)


if __name__ == "__main__":
    main()
