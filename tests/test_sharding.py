"""CPU: tile sharding + the slide-level margin-cell exchange (world_size 2, gloo), and the truth tables
of the per-cell position helpers (cell_detection.py:787-902 restated in cellvit_amd.sharding)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cellvit_amd import sharding as S


def test_shard_tiles_partition():
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 4, 8):
            for block in (1, 4):
                parts = [S.shard_tiles(n, r, world, block) for r in range(world)]
                flat = sorted(i for p in parts for i in p)
                assert flat == list(range(n))
                assert max(len(p) for p in parts) - min(len(p) for p in parts) <= block


def test_global_offset_formula():
    # cell_detection.py:341-350 with patch 1024, overlap 64, downsample 1
    assert S.global_offset(0, 0, 1024, 1, 64) == (-32, -32)
    assert S.global_offset(2, 3, 1024, 1, 64) == (int(2048 - 160), int(3072 - 224))
    assert S.global_offset(1, 0, 1024, 2.0, 64) == (2048 - 96, -32)


def test_cell_status_truth_table():
    def bb(r0, c0, r1, c1):
        return np.array([[r0, c0], [r1, c1]])
    assert S.cell_status(bb(100, 100, 200, 200)) == 0
    assert S.cell_status(bb(64, 64, 960, 960)) == 0            # boundaries are exclusive (< margin, > size-margin)
    assert S.cell_status(bb(10, 10, 30, 30)) == 1
    assert S.cell_status(bb(10, 500, 30, 520)) == 2
    assert S.cell_status(bb(10, 990, 30, 1010)) == 3
    assert S.cell_status(bb(500, 990, 520, 1010)) == 4
    assert S.cell_status(bb(990, 990, 1010, 1010)) == 5
    assert S.cell_status(bb(990, 500, 1010, 520)) == 6
    assert S.cell_status(bb(990, 10, 1010, 30)) == 7
    assert S.cell_status(bb(500, 10, 520, 30)) == 8
    assert S.cell_edge_position(bb(0, 5, 9, 1024)) == [1, 1, 0, 0]
    assert S.cell_edge_position(bb(5, 0, 1024, 9)) == [0, 0, 1, 1]
    assert S.edge_patches([1, 1, 0, 0], 4, 7) == [[3, 7], [3, 8], [4, 8]]
    assert S.edge_patches([0, 0, 0, 1], 4, 7) == [[4, 6]]
    assert S.edge_patches([1, 0, 1, 0], 4, 7) is None            # reference falls through -> None


def _fake_tile(seed):
    rng = np.random.default_rng(seed)
    d = {}
    for i in range(1, 6 + seed):
        r0, c0 = rng.integers(0, 1000, 2)
        d[i] = {"bbox": np.array([[r0, c0], [r0 + 20, c0 + 20]]), "centroid": rng.random(2) * 1024,
                "contour": rng.integers(0, 1024, (3 + i, 2)).astype(np.int32), "type_prob": float(rng.random()),
                "type": int(rng.integers(1, 6))}
    return d


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiles = S.shard_tiles(6, rank, world)
    irs, frs, cts = [], [], []
    for t in tiles:
        ir, fr, ct = S.pack_margin_records(_fake_tile(t), row=t // 3, col=t % 3)
        irs.append(ir); frs.append(fr); cts.append(ct)
    ir, fr, ct = np.concatenate(irs), np.concatenate(frs), np.concatenate(cts)
    gi, gf, gc = S.all_gather_margin_records(ir, fr, ct)
    q.put((rank, gi, gf, gc))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_margin_records_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # expected: rank-ordered concatenation of what each rank packed
    exp_i, exp_f, exp_c = [], [], []
    for rank in range(2):
        for t in S.shard_tiles(6, rank, 2):
            ir, fr, ct = S.pack_margin_records(_fake_tile(t), row=t // 3, col=t % 3)
            exp_i.append(ir); exp_f.append(fr); exp_c.append(ct)
    exp_i, exp_f, exp_c = np.concatenate(exp_i), np.concatenate(exp_f), np.concatenate(exp_c)
    for _, gi, gf, gc in res:
        assert np.array_equal(gi, exp_i) and np.array_equal(gf, exp_f) and np.array_equal(gc, exp_c)
    assert exp_i[:, S.I_CLEN].sum() == len(exp_c)
