"""GPU: the slide-level exchange primitives (cellvit_amd.sharding: all-gatherv of the margin records, gatherv of the writer's
chunks; reference: the single-process CellPostProcessor, cell_detection.py:600-767, has no counterpart) executed by RCCL ITSELF
— backend "nccl", device exchange buffers — as far as ONE GPU allows: a process group of world size 1.  RCCL refuses two ranks
on one device, so the world-2 runs of the same functions are the gloo tests (tests/test_sharding.py, tests/test_cli.py); what
this test adds is that every dtype / shape / zero-length case the route hands to `dist.all_gather` really is accepted by
ProcessGroupNCCL on ROCm (int32 [n,12], float64 [n,3], int32 [m,2], int64 counts, fp32 token rows, empty contributions).

The group lives in a child process behind a timeout: a box whose RCCL cannot initialise skips (environment), wrong results fail."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from cellvit_amd import sharding as S
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=dev)
        dist.barrier()
    except Exception as e:                     # environment: RCCL does not come up on this box
        print("RCCL_INIT_FAILED", repr(e)); sys.exit(77)
    assert dist.get_backend() == "nccl"
    rng = np.random.default_rng(0)
    for n, m in ((0, 0), (1, 3), (257, 4099)):
        ir = rng.integers(-5, 2000, (n, S.N_ICOL)).astype(np.int32)
        fr = rng.random((n, S.N_FCOL))
        ct = rng.integers(0, 1024, (m, 2)).astype(np.int32)
        # the private all-gatherv / gatherv (the public wrappers return early for a world of one): RCCL executes them
        for a in (ir, fr, ct):
            t = torch.from_numpy(a).to(dev)
            parts = S._all_gather_var(t)
            assert len(parts) == 1 and parts[0].is_cuda and parts[0].dtype == t.dtype and torch.equal(parts[0], t), (a.dtype, a.shape)
            got, sent, recv = S._gather_var_to(t, 0)
            assert len(got) == 1 and torch.equal(got[0], t) and sent == 0 and recv == 0
    rows = torch.randn(1000, 1280, device=dev)             # token rows of the writer's gather
    assert torch.equal(torch.cat(S._all_gather_var(rows)), rows)
    t = torch.tensor([41], dtype=torch.int64, device=dev)  # all_gather_int's payload
    out = [torch.zeros_like(t)]
    dist.all_gather(out, t)
    assert int(out[0].item()) == 41
    v = torch.tensor([3.5], dtype=torch.float64, device=dev)   # bench.py's max-over-ranks of the timed region
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    assert float(v.item()) == 3.5
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK")
""")


@pytest.mark.gpu
def test_exchange_primitives_under_rccl_world1(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "child.py"
    script.write_text(CHILD % {"root": ROOT, "port": port})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    try:
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL process group of one rank did not come up within 300 s on this box")
    if r.returncode == 77:
        pytest.skip("RCCL does not initialise on this box: " + r.stdout.strip()[-300:])
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
