"""Generate tests/golden/forward_*.npz by running the IMPORTED REFERENCE (development container
only; /root/reference never travels to the GPU box) on the seeded weights of
cellvit_amd.weights and seeded synthetic tiles.  Commit the outputs together with this script.

    python tools/make_golden_forward.py            # all cases
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import ref_import  # noqa: E402
from cellvit_amd.spec import cellvit256_config, cellvit_sam_config  # noqa: E402
from cellvit_amd.weights import make_state_dict, normalize_tile, synthetic_tile_u8  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def make_input(batch, h, w, first_tile=0):
    xs = []
    for b in range(batch):
        t = synthetic_tile_u8(first_tile + b, size=max(h, w), he_like=(b % 2 == 1))[:h, :w]
        xs.append(normalize_tile(t))
    return torch.from_numpy(np.stack(xs))


def run(name, model, cfg, batch, h, w, full=True, crop=64, tok_crop=None):
    sd = make_state_dict(cfg, seed=0)
    print(name, model.load_state_dict(sd))
    model.eval()
    x = make_input(batch, h, w)
    t = time.time()
    with torch.no_grad():
        out = model(x, retrieve_tokens=True)
    print(f"  reference forward {time.time() - t:.1f}s")
    rec = {"meta_shape": np.array([batch, h, w]), "tissue_types": out["tissue_types"].numpy()}
    for k in ("nuclei_binary_map", "hv_map", "nuclei_type_map", "tokens", "regression_map"):
        if k not in out:
            continue
        v = out[k].numpy()
        if full:
            rec[k] = v
        else:  # large tile: keep a centre crop + corner crop + global statistics
            H, W = v.shape[-2:]
            c = min(crop if (k != "tokens" or tok_crop is None) else tok_crop, H)
            y0, x0 = (H - c) // 2, (W - c) // 2
            rec[k + "_center"] = v[..., y0:y0 + c, x0:x0 + c].copy()
            rec[k + "_corner"] = v[..., :c, :c].copy()
            rec[k + "_stats"] = np.array([v.mean(), v.std(), v.min(), v.max(),
                                          np.abs(v).mean()], dtype=np.float64)
            if k != "tokens":
                rec[k + "_argmax_hist"] = np.bincount(v.argmax(1).ravel(), minlength=v.shape[1])
    np.savez_compressed(os.path.join(OUT, f"forward_{name}.npz"), **rec)


def main():
    os.makedirs(OUT, exist_ok=True)
    cv = ref_import.import_cellvit()
    which = sys.argv[1:] or ["vit256_256", "vit256_b2_128x192", "samb_128", "samh_256", "samh_1024", "vit256_1024",
                             "vit256_nohead_64", "samb_nohead_64", "vit256_reg_64", "samb_reg_64", "vitgen768_64", "saml_64"]
    if "vit256_256" in which:   # BASELINE.json configs[0]
        run("vit256_256", cv.CellViT256(None, 6, 19), cellvit256_config(), 1, 256, 256)
    if "vit256_b2_128x192" in which:   # batch > 1, non-square (bicubic pos-embed w/h handling)
        run("vit256_b2_128x192", cv.CellViT256(None, 6, 19), cellvit256_config(), 2, 128, 192)
    if "samb_128" in which:     # small SAM: window padding (8 -> 14) + interpolated global rel-pos
        run("samb_128", cv.CellViTSAM(None, 6, 19, "SAM-B"), cellvit_sam_config("SAM-B"), 2, 128, 128)
    if "samh_256" in which:
        run("samh_256", cv.CellViTSAM(None, 6, 19, "SAM-H"), cellvit_sam_config("SAM-H"), 1, 256, 256)
    if "samh_1024" in which:    # BASELINE.json configs[2] shape; crops + statistics only
        run("samh_1024", cv.CellViTSAM(None, 6, 19, "SAM-H"), cellvit_sam_config("SAM-H"), 1, 1024, 1024,
            full=False)
    if "vit256_1024" in which:  # BASELINE.json configs[1] shape (4097 tokens); crops + statistics only
        run("vit256_1024", cv.CellViT256(None, 6, 19), cellvit256_config(), 1, 1024, 1024, full=False, tok_crop=16)
    if "vit256_nohead_64" in which:   # num_tissue_classes = 0: head = nn.Identity, tissue_types = norm(x)[:, 0]
        run("vit256_nohead_64", cv.CellViT256(None, 6, 0), cellvit256_config(6, 0), 2, 64, 64)
    if "samb_nohead_64" in which:     # num_tissue_classes = 0: tissue_types = mean of the neck output
        run("samb_nohead_64", cv.CellViTSAM(None, 6, 0, "SAM-B"), cellvit_sam_config("SAM-B", 6, 0), 2, 64, 64)
    if "vit256_reg_64" in which:      # regression_loss=True: binary branch has 4 channels, split 2 + 2 (cellvit.py:191-196)
        run("vit256_reg_64", cv.CellViT256(None, 6, 19, regression_loss=True), cellvit256_config(6, 19, True), 2, 64, 64)
    if "samb_reg_64" in which:        # the same through CellViTSAM.forward (cellvit.py:623-630)
        run("samb_reg_64", cv.CellViTSAM(None, 6, 19, "SAM-B", regression_loss=True), cellvit_sam_config("SAM-B", 6, 19, True), 2, 64, 64)
    if "vitgen768_64" in which:       # the generic class CellViT(...) (cellvit.py:57-75; arch "CellViT" of cell_detection.py:173-190) with non-preset dims:
        from cellvit_amd.spec import cellvit_generic_config   # ViT-B/16 geometry: D 768, 12 heads (hd 64), skip dims 512 / 256, bottleneck 512
        run("vitgen768_64", cv.CellViT(6, 19, 768, 3, 12, 12, [3, 6, 9, 12]), cellvit_generic_config(6, 19, 768, 12, 12, (3, 6, 9, 12)), 2, 64, 64)
    if "saml_64" in which:            # CellViTSAM(..., "SAM-L") (cellvit.py:653-658): D 1024, depth 24, hd 64, global blocks 5 / 11 / 17 / 23
        run("saml_64", cv.CellViTSAM(None, 6, 19, "SAM-L"), cellvit_sam_config("SAM-L"), 2, 64, 64)


if __name__ == "__main__":
    main()
