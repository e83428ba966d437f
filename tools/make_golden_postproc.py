"""Generate tests/golden/postproc_*.npz (development container only).

The reference post-processing (post_proc_cellvit.py:155-249) cannot be imported here (cv2 is not
installable), but its watershed primitive, scikit-image, exists under /opt/conda/bin/python3.9
(skimage 0.18.3; reference pins 0.19.3 — same _watershed_cy algorithm).  For seeded synthetic tiles
this script runs the oracle's stages up to (dist, marker, mask), then runs
skimage.segmentation.watershed(dist, markers=marker, mask=blb) as a black box in the py3.9
interpreter and stores its output, plus the oracle's full per-tile result for cross-machine pinning.

    python tools/make_golden_postproc.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from cellvit_amd.synth import synth_nuclei_maps  # noqa: E402
from oracle import postproc_ref as P  # noqa: E402

PY39 = "/opt/conda/bin/python3.9"
OUT = os.path.join(ROOT, "tests", "golden")

CASES = [  # (name, tile_idx, size, n_cells per 1024^2, magnification)
    ("t0_256_k800", 0, 256, 800, 40),
    ("t1_256_k1500", 1, 256, 1500, 40),
    ("t2_384_k300", 2, 384, 300, 40),
    ("t3_256_k800_x20", 3, 256, 800, 20),
    ("t4_512_k1200", 4, 512, 1200, 40),
]

SK = r"""
import sys, numpy as np
from skimage.segmentation import watershed
d = np.load(sys.argv[1])
out = watershed(d["dist"], markers=d["marker"], mask=d["blb"])
np.save(sys.argv[2], out.astype(np.int32))
"""


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, idx, size, k, mag in CASES:
        tm, bm, hv, _ = synth_nuclei_maps(idx, size, k)
        obj, ks = (10, 21) if mag == 40 else (3, 11)
        inst, blb, dist, marker = P.proc_np_hv(bm, hv[0], hv[1], obj, ks, debug=True)
        with tempfile.TemporaryDirectory() as td:
            np.savez(os.path.join(td, "in.npz"), dist=dist, marker=marker, blb=blb)
            subprocess.run([PY39, "-c", SK, os.path.join(td, "in.npz"), os.path.join(td, "out.npy")], check=True,
                           stderr=subprocess.DEVNULL)
            sk = np.load(os.path.join(td, "out.npy"))
        print(name, "oracle == skimage:", bool((sk == inst).all()), "instances", len(np.unique(inst)) - 1,
              "fg", float(bm.mean()))
        pm = np.stack([tm.astype(np.float32), bm.astype(np.float32), hv[0], hv[1]], -1)
        inst2, d = P.postprocess_tile(pm, 6, mag)
        assert (inst2 == inst).all()
        ids = np.array(sorted(d.keys()), dtype=np.int32)
        np.savez_compressed(
            os.path.join(OUT, f"postproc_{name}.npz"),
            meta=np.array([idx, size, k, mag]), skimage_watershed=sk, oracle_inst=inst,
            ids=ids, bbox=np.array([d[i]["bbox"].ravel() for i in ids], dtype=np.int32),
            centroid=np.array([d[i]["centroid"] for i in ids], dtype=np.float64),
            type=np.array([d[i]["type"] for i in ids], dtype=np.int32),
            type_prob=np.array([d[i]["type_prob"] for i in ids], dtype=np.float64),
            contour_len=np.array([len(d[i]["contour"]) for i in ids], dtype=np.int32),
            contour_cat=np.concatenate([d[i]["contour"] for i in ids]).astype(np.int32) if len(ids) else np.zeros((0, 2), np.int32))


if __name__ == "__main__":
    main()
