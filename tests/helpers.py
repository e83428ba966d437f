"""Shared helpers for the parity tests (seeded inputs identical to tools/make_golden_forward.py)."""
import os

import numpy as np
import torch

from cellvit_amd.spec import cellvit256_config, cellvit_generic_config, cellvit_sam_config
from cellvit_amd.weights import make_state_dict, normalize_tile, synthetic_tile_u8

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    # name: (config factory, batch, H, W)
    "vit256_256": (lambda: cellvit256_config(), 1, 256, 256),
    "vit256_b2_128x192": (lambda: cellvit256_config(), 2, 128, 192),
    "samb_128": (lambda: cellvit_sam_config("SAM-B"), 2, 128, 128),
    "samh_256": (lambda: cellvit_sam_config("SAM-H"), 1, 256, 256),
    "samh_1024": (lambda: cellvit_sam_config("SAM-H"), 1, 1024, 1024),
    "vit256_1024": (lambda: cellvit256_config(), 1, 1024, 1024),          # BASELINE.json configs[1] geometry (4097 tokens)
    "vit256_nohead_64": (lambda: cellvit256_config(6, 0), 2, 64, 64),     # num_tissue_classes = 0: head = nn.Identity
    "samb_nohead_64": (lambda: cellvit_sam_config("SAM-B", 6, 0), 2, 64, 64),
    "vit256_reg_64": (lambda: cellvit256_config(6, 19, True), 2, 64, 64),    # regression_loss=True (cellvit.py:191-196)
    "samb_reg_64": (lambda: cellvit_sam_config("SAM-B", 6, 19, True), 2, 64, 64),
    "vitgen768_64": (lambda: cellvit_generic_config(6, 19, 768, 12, 12, (3, 6, 9, 12)), 2, 64, 64),   # generic CellViT(...), cellvit.py:57-75
    "saml_64": (lambda: cellvit_sam_config("SAM-L"), 2, 64, 64),                                       # CellViTSAM(..., "SAM-L"), cellvit.py:653-658
}


def make_input(batch, h, w, first_tile=0):
    xs = []
    for b in range(batch):
        t = synthetic_tile_u8(first_tile + b, size=max(h, w), he_like=(b % 2 == 1))[:h, :w]
        xs.append(normalize_tile(t))
    return torch.from_numpy(np.stack(xs))


def load_case(name):
    mk, b, h, w = CASES[name]
    cfg = mk()
    sd = make_state_dict(cfg, seed=0)
    x = make_input(b, h, w)
    gold = dict(np.load(os.path.join(GOLDEN, f"forward_{name}.npz")))
    return cfg, sd, x, gold


def compare_outputs(out, gold, atol, rtol=0.0, keys=("tissue_types", "nuclei_binary_map", "hv_map",
                                                      "nuclei_type_map", "tokens"), atol_tokens=None):
    """Return dict key -> max abs error; assert within tolerance.  `atol_tokens`: separate bound for the raw encoder
    tokens (abs max ~25 for SAM-H, two orders above the logits)."""
    errs = {}
    atol_all = atol
    for k in keys:
        atol = atol_tokens if (k == "tokens" and atol_tokens is not None) else atol_all
        a = out[k].detach().float().cpu().numpy()
        if k in gold:
            g = gold[k]
            assert a.shape == g.shape, (k, a.shape, g.shape)
            e = float(np.abs(a - g).max())
            errs[k] = e
            assert np.allclose(a, g, atol=atol, rtol=rtol), f"{k}: max abs err {e} > {atol}"
        elif k + "_center" in gold:
            gc = gold[k + "_center"]
            c = gc.shape[-1]
            H, W = a.shape[-2:]
            y0, x0 = (H - c) // 2, (W - c) // 2
            e1 = float(np.abs(a[..., y0:y0 + c, x0:x0 + c] - gc).max())
            e2 = float(np.abs(a[..., :c, :c] - gold[k + "_corner"]).max())
            errs[k] = max(e1, e2)
            assert errs[k] <= atol, f"{k}: crop max abs err {errs[k]} > {atol}"
    return errs
