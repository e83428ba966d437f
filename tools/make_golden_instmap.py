"""Generate tests/golden/instmap_cases.npz: inputs + outputs of the REFERENCE `CellViT.generate_instance_nuclei_map`
(models/segmentation/cell_segmentation/cellvit.py:385-414), imported here in the dev container with the stubs of
tools/ref_import.py.  The reference never travels: only this data file does."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from cellvit_amd.synth import synth_nuclei_maps  # noqa: E402
from ref_import import import_cellvit  # noqa: E402


def main():
    ref = import_cellvit()
    rng = np.random.default_rng(11)
    B, T = 3, 256
    inst = np.zeros((B, T, T), np.int32)
    ids_all, types_all = [], []
    for b in range(B):
        _, _, _, gt = synth_nuclei_maps(300 + b, T, 900 + 300 * b)
        gt = gt.astype(np.int32)
        gt[gt > 0] = gt[gt > 0] * 3 + b                     # ids with gaps, as the watershed leaves them
        inst[b] = gt
        ids = np.array([i for i in np.unique(gt) if i], np.int32)
        ty = rng.integers(0, 6, len(ids)).astype(np.int32)   # type 0 cells stay in the dict (reference paints channel 0)
        drop = rng.random(len(ids)) < 0.15                   # quirk 4: instances absent from the dict are never painted
        ids_all.append(ids[~drop]); types_all.append(ty[~drop])
    type_preds = [{int(i): {"type": int(t)} for i, t in zip(ids_all[b], types_all[b])} for b in range(B)]
    stub = types.SimpleNamespace(num_nuclei_classes=6)
    out = ref.CellViT.generate_instance_nuclei_map(stub, torch.from_numpy(inst).float(), type_preds)
    data = {"inst": inst, "out": out.numpy().astype(np.float32), "n": np.array(B)}
    for b in range(B):
        data[f"ids{b}"] = ids_all[b]; data[f"types{b}"] = types_all[b]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "instmap_cases.npz"), **data)
    print("saved", out.shape, out.dtype, float(out.max()))


if __name__ == "__main__":
    main()
