"""Micro-benchmark of the attention op entry (qkv projection + attention core) on the SAM-H global-block shape; the attention
core's share is read from the difference between kernel variants (CVA_ATTN / CVA_ATTN3_DBG, ablation flavour: CVA_LIB=abl).
    CVA_LIB=abl CVA_ATTN=4 python tools/bench_attn.py [B gh gw heads D win] [iters]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cellvit_amd import _lib  # noqa: E402


def main():
    B, gh, gw, heads, D, win = (int(a) for a in sys.argv[1:7]) if len(sys.argv) >= 7 else (16, 64, 64, 16, 1280, 0)
    iters = int(sys.argv[7]) if len(sys.argv) >= 8 else 5
    lib = _lib.load()
    hd = D // heads
    ntok = gh * gw
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B * ntok, D, device="cuda", generator=g).half()
    W = (torch.randn(3 * D, D, device="cuda", generator=g) / D ** 0.5).half()
    b = torch.randn(3 * D, device="cuda", generator=g) * 0.3
    KH, KW = (win, win) if win else (gh, gw)
    th = torch.randn(2 * KH - 1, hd, device="cuda", generator=g) * 0.2
    tw = torch.randn(2 * KW - 1, hd, device="cuda", generator=g) * 0.2
    out = torch.zeros(B * ntok, D, device="cuda", dtype=torch.float16)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    call = lambda: lib.cv_op_attention(0, p(x), p(W), p(b), p(th), p(tw), p(out), B, gh, gw, 0, heads, D, win, None)  # noqa: E731
    for _ in range(2):
        _lib.check(call())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * B * ntok * (win * win if win else ntok) * hd * heads
    print(f"CVA_ATTN={os.environ.get('CVA_ATTN', '-')} CVA_ATTN3_DBG={os.environ.get('CVA_ATTN3_DBG', '-')} B={B} {gh}x{gw} heads={heads} D={D} win={win}: "
          f"{ms * 1e3:.1f} us per call (qkv projection + scratch allocation + attention); attention core FLOPs {fl / 1e9:.1f} G; checksum {float(out.float().abs().mean()):.6f}")


if __name__ == "__main__":
    main()
