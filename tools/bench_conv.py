"""Micro-benchmark of the NHWC 3x3 convolution entry point on the decoder's layer shapes (B = 8 tiles of 1024^2).
    python tools/bench_conv.py [iters]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cellvit_amd import _lib  # noqa: E402

SHAPES_ALL = [  # (H, C1, C2, Cout)
    (128, 512, 512, 512), (128, 512, 0, 512), (256, 256, 256, 256), (256, 256, 0, 256),
    (512, 128, 128, 128), (512, 128, 0, 128), (1024, 64, 64, 64), (1024, 64, 0, 64), (1024, 32, 0, 64),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    sel = os.environ.get("CONV_SHAPES")          # e.g. CONV_SHAPES=7,5,1
    SHAPES = [SHAPES_ALL[int(i)] for i in sel.split(",")] if sel else SHAPES_ALL[:8]
    B = 8
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    tot = 0.0
    for H, C1, C2, Co in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(0)
        s1 = torch.randn(B, H, H, C1, device="cuda", generator=g).half()
        s2 = torch.randn(B, H, H, C2, device="cuda", generator=g).half() if C2 else None
        W = (torch.randn(Co, 9 * (C1 + C2), device="cuda", generator=g) / (9 * (C1 + C2)) ** 0.5).half()
        b = torch.zeros(Co, device="cuda")
        out = torch.empty(B, H, H, Co, device="cuda", dtype=torch.float16)
        for _ in range(2):
            _lib.check(lib.cv_op_conv3x3(0, p(s1), C1, p(s2), C2, p(W), p(b), p(out), 0, B, H, H, Co, 1, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.cv_op_conv3x3(0, p(s1), C1, p(s2), C2, p(W), p(b), p(out), 0, B, H, H, Co, 1, None)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * B * H * H * 9 * (C1 + C2) * Co
        tot += ms
        print(f"H={H:5d} Cin={C1}+{C2:<4d} Cout={Co:4d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.0f} TFLOP/s   out {B * H * H * Co * 2 / 1e6:.0f} MB")
    print(f"sum {tot:.2f} ms")


if __name__ == "__main__":
    main()
