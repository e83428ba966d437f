"""LayerNorm micro-benchmark (rows x 1280 fp32 -> fp16): python tools/bench_ln.py [rows] [iters]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cellvit_amd import _lib  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    Cc = 1280
    lib = _lib.load()
    x = torch.randn(M, Cc, device="cuda")
    g = torch.rand(Cc, device="cuda") + 0.5
    b = torch.randn(Cc, device="cuda") * 0.1
    out = torch.empty(M, Cc, device="cuda", dtype=torch.float16)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for _ in range(3):
        _lib.check(lib.cv_op_layernorm(0, p(x), p(g), p(b), p(out), 0, M, Cc, C.c_float(1e-6), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.cv_op_layernorm(0, p(x), p(g), p(b), p(out), 0, M, Cc, C.c_float(1e-6), None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    ref = torch.nn.functional.layer_norm(x[:1024], (Cc,), g, b, 1e-6)
    err = float((out[:1024].float() - ref).abs().max())
    print(f"LN {M}x{Cc}: {ms * 1e3:.1f} us  {M * Cc * 6 / ms / 1e9:.2f} TB/s  maxerr {err:.2e}  (CVA_LN={os.environ.get('CVA_LN', '0')})")


if __name__ == "__main__":
    main()
