"""Throughput of the inference CLI's tile loop (cellvit_amd.inference.cell_detection.run_tiles) on a synthetic pre-patched
slide: N tiles of 1024^2 (two distinct PNGs, repeated), CellViT-SAM-H fp16 with seeded random weights.  Measures what the
reference's DataLoader + tile loop does (cell_detection.py:266-421): PNG decode in worker threads -> pinned u8 batches ->
forward (inference transform fused) -> on-device post-processing -> on-device token pooling -> record arrays on the host.

    python tools/bench_cli.py [--tiles 128] [--batch 16] [--workers N] [--model samh|vit256]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=128)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--model", default="samh", choices=["samh", "vit256"])
    args = ap.parse_args()
    import numpy as np
    import torch
    import yaml
    from PIL import Image
    from cellvit_amd.inference import cell_detection as CD
    from cellvit_amd.spec import cellvit256_config, cellvit_sam_config
    from cellvit_amd.weights import make_state_dict, synthetic_tile_u8

    tmp = tempfile.mkdtemp(prefix="cva_cli_")
    cfg = cellvit_sam_config("SAM-H") if args.model == "samh" else cellvit256_config()
    ckpt = {"arch": "CellViTSAM" if args.model == "samh" else "CellViT256", "model_state_dict": make_state_dict(cfg, 0),
            "config": {"data.num_nuclei_classes": 6, "data.num_tissue_classes": 19, "model.backbone": "SAM-H" if args.model == "samh" else "default",
                       "training.mixed_precision": True,
                       "dataset_config.nuclei_types": {"Background": 0, "Neoplastic": 1, "Inflammatory": 2, "Connective": 3,
                                                        "Dead": 4, "Epithelial": 5}}}
    torch.save(ckpt, os.path.join(tmp, "ckpt.pth"))
    slide = os.path.join(tmp, "slide")
    os.makedirs(os.path.join(slide, "patches"))
    for i in range(2):
        Image.fromarray(synthetic_tile_u8(i, 1024, he_like=True)).save(os.path.join(slide, "patches", f"src{i}.png"))
    side = int(np.ceil(np.sqrt(args.tiles)))
    meta = []
    for t in range(args.tiles):
        row, col = divmod(t, side)
        name = f"slide_{row}_{col}.png"
        os.symlink(f"src{t % 2}.png", os.path.join(slide, "patches", name))
        meta.append({name: {"row": row, "col": col}})
    with open(os.path.join(slide, "patch_metadata.json"), "w") as f:
        json.dump(meta, f)
    with open(os.path.join(slide, "metadata.yaml"), "w") as f:
        yaml.safe_dump({"magnification": 40, "downsampling": 1, "patch_size": 1024, "patch_overlap": 64,
                        "label_map": {"background": 0}, "base_magnification": 40}, f)
    inf = CD.CellSegmentationInference(os.path.join(tmp, "ckpt.pth"), 0)
    wsi = CD.PatchedSlide("slide", slide)
    ids = list(range(args.tiles))
    inf.run_tiles(wsi, ids[:2 * args.batch], args.batch, num_workers=args.workers)          # warm-up (engine, workspaces)
    torch.cuda.synchronize()
    # decode-only rate of the prefetcher (no GPU work)
    t0 = time.perf_counter()
    n = 0
    for b_ids, x_u8, _ in CD.TilePrefetcher(wsi, ids, args.batch, inf.device, args.workers):
        n += len(b_ids)
    torch.cuda.synchronize()
    t_dec = time.perf_counter() - t0
    t0 = time.perf_counter()
    local, processed, stats = inf.run_tiles(wsi, ids, args.batch, num_workers=args.workers)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"tool": "bench_cli", "model": args.model, "tiles": args.tiles, "batch": args.batch,
                      "workers": args.workers, "host_cpus": os.cpu_count(),
                      "decode_only_tiles_per_s": n / t_dec, "tile_loop_tiles_per_s": args.tiles / dt, "tile_loop_s": dt,
                      "cells": len(local)}))


if __name__ == "__main__":
    main()
