"""Micro-benchmark of one attention layer through cv_op_attention (run under rocprofv3 --kernel-trace --stats to
split the qkv GEMM / pad / attention kernels).   python tools/bench_attn.py B win rel [iters]"""
import ctypes as C
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cellvit_amd import _lib  # noqa: E402


def main():
    B, win, rel = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    gh = gw = 64
    heads, D = 16, 1280
    hd = D // heads
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B * gh * gw, D, device="cuda", generator=g).half()
    W = (torch.randn(3 * D, D, device="cuda", generator=g) / math.sqrt(D)).half()
    b = torch.randn(3 * D, device="cuda", generator=g) * 0.3
    KH = win if win else gh
    th = torch.randn(2 * KH - 1, hd, device="cuda", generator=g) * 0.2
    tw = torch.randn(2 * KH - 1, hd, device="cuda", generator=g) * 0.2
    out = torch.zeros(B * gh * gw, D, device="cuda", dtype=torch.float16)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(iters + 2):
        if it == 2:
            e0.record()
        _lib.check(lib.cv_op_attention(0, p(x), p(W), p(b), p(th) if rel else None, p(tw) if rel else None, p(out),
                                       B, gh, gw, 0, heads, D, win, None))
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} win={win} rel={rel}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per layer (qkv + pad + attention)")


if __name__ == "__main__":
    main()
