"""Post-processing timing on synthetic nucleus maps (B tiles of 1024^2): python tools/bench_pp.py [B] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cellvit_amd.postproc import postprocess_device  # noqa: E402
from cellvit_amd.synth import synth_nuclei_maps  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    maps = [synth_nuclei_maps(i, 1024, 800) for i in range(B)]
    t = torch.from_numpy(np.stack([m[0] for m in maps])).cuda()
    b = torch.from_numpy(np.stack([m[1] for m in maps])).cuda()
    hv = torch.from_numpy(np.stack([m[2] for m in maps])).cuda()
    for _ in range(2):
        res = postprocess_device(b, t, hv, 6, 10, 21, want_contours=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        res = postprocess_device(b, t, hv, 6, 10, 21, want_contours=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B}: {e0.elapsed_time(e1) / iters:.3f} ms per batch, instances {int(res[2].sum())}")


if __name__ == "__main__":
    main()
