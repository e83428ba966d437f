"""Micro-benchmark of the ConvTranspose2d(k2, s2) entry point on the decoder's layer shapes (B = 8 tiles of 1024^2).
    python tools/bench_convt.py [iters]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cellvit_amd import _lib  # noqa: E402

SHAPES = [(64, 1280, 512), (128, 512, 256), (256, 256, 128), (512, 128, 64), (64, 1280, 256), (128, 512, 512)]   # (H in, Cin, Cout)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    B = int(os.environ.get("B", "8"))
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    tot = 0.0
    for H, Ci, Co in SHAPES:
        x = torch.randn(B, H, H, Ci, device="cuda").half()
        W = (torch.randn(4 * Co, Ci, device="cuda") / Ci ** 0.5).half()
        b4 = torch.zeros(4 * Co, device="cuda")
        out = torch.empty(B, 2 * H, 2 * H, Co, device="cuda", dtype=torch.float16)
        for _ in range(2):
            _lib.check(lib.cv_op_convT2x2(0, p(x), p(W), p(b4), p(out), B, H, H, Ci, Co, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.cv_op_convT2x2(0, p(x), p(W), p(b4), p(out), B, H, H, Ci, Co, None)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * B * H * H * 4 * Co * Ci
        by = x.numel() * 2 + out.numel() * 2
        tot += ms
        print(f"H={H:4d} Cin={Ci:5d} Cout={Co:4d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.0f} TFLOP/s  {by / ms / 1e9:6.2f} TB/s (in {x.numel() * 2 / 1e6:.0f} MB + out {out.numel() * 2 / 1e6:.0f} MB)")
    print(f"sum {tot:.2f} ms")


if __name__ == "__main__":
    main()
