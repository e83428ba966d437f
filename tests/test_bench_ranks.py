"""bench.py's multi-rank path (BASELINE.json configs[3] / ⑤): `--gpus N` runs one rank per GPU over RCCL and, after the timed tile loop, the slide-level route
on the SAME process group.  A 1-GPU box cannot run two RCCL ranks (one device per rank), so the GPU test below drives the identical code path — spawn, barrier /
max-over-ranks timing, per-rank clocks, the slide leg with its margin-record all-gatherv and the writer's point-to-point gather — with two ranks sharing cuda:0 over gloo
(`--backend gloo --share-gpu`, both recorded in the line's config)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_bench_share_gpu_needs_gloo():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--backend gloo" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_run_the_slide_leg_on_their_process_group():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--batch", "8", "--steps", "1", "--warmup", "1",
           "--slide-tiles-per-rank", "16", "--no-kernel-events"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["ranks"] == 2 and rec["config"]["collective_backend"] == "gloo"
    assert len(rec["config"]["per_rank_tiles_per_s"]) == 2 and "share_gpu" in rec["config"]["experiment_env"]
    assert rec["value"] > 0 and rec["scaling"] == "weak"
    s = rec["extra"]["slide"]
    assert s["ranks"] == 2 and s["tiles"] == 32 and s["collective_backend"] == "gloo" and s["exchange_buffers"] == "host"
    assert s["cells_written"] > 0 and s["margin_records"] > s["margin_kept"] > 0
    assert s["writer_gather_bytes_received_rank0"] > 0 and s["exchange_s"] >= 0
