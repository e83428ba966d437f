"""Slide-level benchmark (BASELINE.json configs[3]): a synthetic pre-patched slide of N 1024^2 tiles with 64-px overlap
through the whole CLI route — PNG decode -> forward (CellViT-SAM-H fp16, seeded random weights, really executed) ->
on-device post-processing -> token pooling -> records -> margin-record exchange -> slide-level de-duplication -> writers.

Random-weight logits contain no nuclei, so after every forward the argmax planes / HV map handed to the post-processing are
REPLACED by the tile's crop of a periodic synthetic nucleus world (cellvit_amd.synth.synth_world_maps, ~800 nuclei per tile):
neighbouring tiles then see the same nuclei in their overlap, which is what the stitch needs as input.  The forward's cost is
in the tile loop's time; its outputs are not used.

    python tools/bench_slide.py [--tiles 1024] [--batch 16] [--model samh|vit256] [--ranks N] [--backend nccl|gloo]
`--ranks N --backend nccl` (the default backend when the box has >= N GPUs): one process per GPU, exchange buffers on the device,
margin-record all-gatherv and the writer's point-to-point gather over RCCL/xGMI — the configs[3] route as the CLI runs it.
`--ranks 2 --backend gloo` on a 1-GPU box: two processes share cuda:0 and exchange through host buffers — it exercises the sharded
route (margin-record all-gatherv, replicated stitch, writer's gather), not a scaling claim.
Prints one JSON line: tile-loop tiles/s, seconds of exchange / stitch / to_dicts / writers, cell counts.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class WorldMapsModel:
    """The real model, whose planes / HV outputs are replaced by world crops selected by pixel (0, 0, 0) of each tile."""

    def __init__(self, real, world, device):
        import numpy as np
        import torch
        from cellvit_amd.synth import world_tile
        self.real = real
        self.patch_size, self.num_nuclei_classes, self.embed_dim = real.patch_size, real.num_nuclei_classes, real.embed_dim
        crops = [world_tile(world, r, c) for r in range(2) for c in range(2)]           # period 2 tiles in each direction
        self.typ = torch.from_numpy(np.stack([c[0] for c in crops])).to(device)
        self.bin = torch.from_numpy(np.stack([c[1] for c in crops])).to(device)
        self.hv = torch.from_numpy(np.stack([c[2] for c in crops])).to(device)
        self._last_argmax = None

    @property
    def _last_engine(self):           # (the tile loop releases the previous batch's post-processing by the engine's stage event)
        return self.real._last_engine

    def forward_u8(self, x_u8, mean, std, retrieve_tokens=False):
        out = self.real.forward_u8(x_u8, mean, std, retrieve_tokens=retrieve_tokens)
        cls = x_u8[:, 0, 0, 0].long()
        self._last_argmax = (self.bin[cls], self.typ[cls])
        out["hv_map"] = self.hv[cls]
        return out


NUCLEI_TYPES = {"Background": 0, "Neoplastic": 1, "Inflammatory": 2, "Connective": 3, "Dead": 4, "Epithelial": 5}


def build_slide(tmp, tiles, model, with_ckpt=True):
    import numpy as np
    import torch
    import yaml
    from PIL import Image
    from cellvit_amd.spec import cellvit256_config, cellvit_sam_config
    from cellvit_amd.weights import make_state_dict, synthetic_tile_u8
    if with_ckpt:
        cfg = cellvit_sam_config("SAM-H") if model == "samh" else cellvit256_config()
        ckpt = {"arch": "CellViTSAM" if model == "samh" else "CellViT256", "model_state_dict": make_state_dict(cfg, 0),
                "config": {"data.num_nuclei_classes": 6, "data.num_tissue_classes": 19, "model.backbone": "SAM-H" if model == "samh" else "default",
                           "training.mixed_precision": True, "dataset_config.nuclei_types": NUCLEI_TYPES}}
        torch.save(ckpt, os.path.join(tmp, "ckpt.pth"))
    slide = os.path.join(tmp, "slide")
    os.makedirs(os.path.join(slide, "patches"))
    for k in range(4):                                   # one source image per (row % 2, col % 2) class; pixel (0,0,0) names the class
        img = synthetic_tile_u8(k, 1024, he_like=True).copy()
        img[0, 0, 0] = k
        Image.fromarray(img).save(os.path.join(slide, "patches", f"src{k}.png"))
    side = int(np.ceil(np.sqrt(tiles)))
    meta = []
    for t in range(tiles):
        row, col = divmod(t, side)
        name = f"slide_{row}_{col}.png"
        os.symlink(f"src{(row % 2) * 2 + (col % 2)}.png", os.path.join(slide, "patches", name))
        meta.append({name: {"row": row, "col": col}})
    with open(os.path.join(slide, "patch_metadata.json"), "w") as f:
        json.dump(meta, f)
    with open(os.path.join(slide, "metadata.yaml"), "w") as f:
        yaml.safe_dump({"magnification": 40, "downsampling": 1, "patch_size": 1024, "patch_overlap": 64,
                        "label_map": {"background": 0}, "base_magnification": 40}, f)
    return os.path.join(tmp, "ckpt.pth"), slide


def run(tiles=1024, batch=16, model="samh", geojson=False, tmp=None, warmup_batches=2, world_seed=3, real_model=None, slides=1, stream_tail=True,
        overlap_postproc=True):
    """One slide through process_wsi on this process's rank; returns the stats dict of rank 0 (None on other ranks).
    `real_model`: an already built cellvit_amd model (bench.py's); otherwise a seeded checkpoint is written and loaded
    through the CLI's own checkpoint path."""
    import torch
    import torch.distributed as dist
    from cellvit_amd.inference import cell_detection as CD
    from cellvit_amd.synth import synth_world_maps
    rank = dist.get_rank() if dist.is_initialized() else 0
    world_n = dist.get_world_size() if dist.is_initialized() else 1
    if world_n > 1:       # one slide directory for all ranks: rank 0's choice travels (a launcher gives every rank its own mkdtemp otherwise)
        box = [tmp if tmp is not None else tempfile.mkdtemp(prefix="cva_slide_")] if rank == 0 else [None]
        dist.broadcast_object_list(box, src=0)
        tmp = box[0]
    elif tmp is None:
        tmp = tempfile.mkdtemp(prefix="cva_slide_")
    if rank == 0:
        build_slide(tmp, tiles, model, with_ckpt=real_model is None)
    if world_n > 1:
        dist.barrier()
    ckpt, slide = os.path.join(tmp, "ckpt.pth"), os.path.join(tmp, "slide")
    if real_model is None:
        inf = CD.CellSegmentationInference(ckpt, torch.cuda.current_device())
    else:                                                # the CLI object around a model that already lives on the device
        import logging
        inf = CD.CellSegmentationInference.__new__(CD.CellSegmentationInference)
        inf.logger = logging.getLogger("cellvit_amd")
        inf.device = torch.device("cuda", torch.cuda.current_device())
        inf.run_conf = {"dataset_config": {"nuclei_types": NUCLEI_TYPES}}
        inf.mixed_precision, inf.model, inf.mean, inf.std, inf.pool_cap = True, real_model, (0.5,) * 3, (0.5,) * 3, 2048
    inf.overlap_postproc = overlap_postproc
    inf.model = WorldMapsModel(inf.model, synth_world_maps(world_seed, 1920, 2800), inf.device)
    wsi = CD.PatchedSlide("slide", slide)
    inf.run_tiles(wsi, list(range(min(tiles, warmup_batches * batch))), batch)               # warm-up (engine, workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = inf.process_wsi(wsi, batch_size=batch, geojson=geojson, stream_tail=stream_tail)
    total = time.perf_counter() - t0
    dataset = None
    if slides > 1:        # process_dataset mode: the files of slide k are written while the tile loop of slide k+1 runs
        t1 = time.perf_counter()
        loops = []
        for k in range(slides):
            st = inf.process_wsi(wsi, subdir_name=f"s{k}", batch_size=batch, geojson=geojson, defer_write=True, stream_tail=stream_tail)
            loops.append(st["tiles"] / st["t_loop"])
        inf.wait_for_writers()
        dt = time.perf_counter() - t1
        dataset = {"slides": slides, "total_s": dt, "s_per_slide": dt / slides, "tiles_per_s": slides * tiles / dt,
                   "tile_loop_tiles_per_s_per_slide": loops}
    out_bytes = sum(os.path.getsize(os.path.join(stats["outdir"], f)) for f in os.listdir(stats["outdir"])) if rank == 0 else 0
    if rank != 0:
        return None
    import hashlib
    digests = {}
    for f in ("cells.json", "cell_detection.json"):          # (outside every timed interval) so that two runs of the same slide can be compared byte for byte
        h = hashlib.sha256()
        with open(os.path.join(stats["outdir"], f), "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 24), b""):
                h.update(blk)
        digests[f] = h.hexdigest()[:16]
    return {"tool": "bench_slide", "model": model, "tiles": tiles, "batch": batch, "ranks": world_n,
            "collective_backend": dist.get_backend() if dist.is_initialized() else None,
            "exchange_buffers": ("device" if dist.get_backend() == "nccl" else "host") if dist.is_initialized() else None,
            "device_of_rank0": torch.cuda.get_device_name(torch.cuda.current_device()) + f" (cuda:{torch.cuda.current_device()})",
            "tile_loop_tiles_per_s_rank0": stats["tiles"] / stats["t_loop"], "tile_loop_s": stats["t_loop"],
            "cells_before_cleaning_rank0": stats["cells_before_cleaning"], "cells_written": stats["n_cells"],
            "margin_records": stats["margin_records"], "margin_kept": stats["margin_kept"],
            "exchange_s": stats["exchange_s"], "stitch_s": stats["stitch_s"], "to_dicts_s": stats["to_dicts_s"],
            "margin_bytes_all_gathered": stats.get("margin_bytes_all_gathered"), "writer_gather_bytes_received_rank0": stats.get("writer_gather_bytes_received"),
            "write_s": stats["write_s"], "slide_total_s": total, "slide_tiles_per_s": tiles / total,
            "postproc_schedule": "second stream, released by the next forward's full-resolution stage event" if overlap_postproc else "back to back on one stream",
            "tail_s": total - stats["t_loop"], "tail_route": "streamed" if stream_tail else "batch", "tail_wait_workers_s": stats.get("tail_wait_workers_s"), "tail_breakdown_s": stats.get("tail_breakdown_s"),
            "output_MB": out_bytes / 1e6, "output_sha256_16": digests, "host_cpus": os.cpu_count(), "dataset_mode": dataset}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--model", default="samh", choices=["samh", "vit256"])
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="process-group backend for --ranks > 1; default: nccl (RCCL, one GPU per rank) when the box has that many GPUs, else gloo "
                         "(all ranks on cuda:0, host exchange buffers)")
    ap.add_argument("--geojson", action="store_true")
    ap.add_argument("--tmp", default=None)
    ap.add_argument("--slides", type=int, default=1, help="> 1: additionally run that many slides back to back with deferred writers")
    ap.add_argument("--serial-postproc", action="store_true", help="forward and post-processing of a batch back to back on one stream (the reference's order)")
    ap.add_argument("--batch-tail", action="store_true", help="the batch route of the slide tail (finalize_slide + write_outputs) instead of the streaming one")
    args = ap.parse_args()
    if args.ranks > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        tmp = tempfile.mkdtemp(prefix="cva_slide_")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + ["--tmp", tmp]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = args.backend or ("nccl" if torch.cuda.device_count() >= world else "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0      # nccl: one GPU per rank (RCCL refuses two ranks on one device)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")               # all ranks share the box's one GPU: host-side exchange
    rec = run(args.tiles, args.batch, args.model, args.geojson, args.tmp, slides=args.slides, stream_tail=not args.batch_tail,
              overlap_postproc=not args.serial_postproc)
    if rec is not None:
        print(json.dumps(rec))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
