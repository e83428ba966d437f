"""Generate tests/golden/eval_cases.npz: inputs + outputs of the REFERENCE evaluation metrics (`binarize`,
`cell_detection_scores`, `cell_type_detection_scores` of `cell_segmentation/utils/metrics.py`, imported here in the dev
container; numpy + scipy only) on seeded inputs.  The reference never travels: only this data file does."""
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


def main():
    rng = np.random.default_rng(77)
    data = {}
    # binarize: per-class instance maps with a few cross-channel overlaps and non-contiguous ids
    for c in range(4):
        types, _, _, inst = synth_nuclei_maps(300 + c, 128, (25 + 6 * c) * 64)
        x = np.zeros((128, 128, 5), np.int32)
        for k in range(5):
            x[:, :, k] = np.where(types == k + 1, inst * (3 if c == 2 else 1), 0)
        if c >= 1:                                   # overlaps: copy a block of one channel's ids into a later channel
            x[20:60, 30:90, 4] = np.where(x[20:60, 30:90, 0] > 0, x[20:60, 30:90, 0] + 1000, x[20:60, 30:90, 4])
        if c == 3:
            x[:, :, 2] = 0                           # an empty channel
        data[f"bin{c}_x"] = x
        data[f"bin{c}_out"] = ref.binarize(x)
    data["n_bin"] = np.array(4)
    # detection scores on random pairings
    for c in range(5):
        n_pair, n_ut, n_up = int(rng.integers(20, 200)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
        pt = rng.integers(0, 6, n_pair); pp = np.where(rng.random(n_pair) < 0.7, pt, rng.integers(0, 6, n_pair))
        ut = rng.integers(0, 6, n_ut); up = rng.integers(0, 6, n_up)
        if c == 4:
            pt[:5] = -1
        data[f"det{c}_pt"], data[f"det{c}_pp"], data[f"det{c}_ut"], data[f"det{c}_up"] = pt, pp, ut, up
        data[f"det{c}_all"] = np.array(ref.cell_detection_scores(pt, pp, ut, up), np.float64)
        data[f"det{c}_type"] = np.array([ref.cell_type_detection_scores(pt, pp, ut, up, t) for t in range(1, 6)], np.float64)
        data[f"det{c}_type_nonex"] = np.array([ref.cell_type_detection_scores(pt, pp, ut, up, t, exhaustive=False)
                                               for t in range(1, 6)], np.float64)
    data["n_det"] = np.array(5)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "eval_cases.npz"), **data)
    print({k: v.shape for k, v in data.items() if k.endswith("_out")})


if __name__ == "__main__":
    main()
