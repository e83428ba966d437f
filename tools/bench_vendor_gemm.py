"""fc1 / fc2 of the SAM-H step on this repository's kernel and on the vendor library torch dispatches to (hipBLASLt), a few launches each —
the workload of the PMC comparison (profiles/scripts/gpu_r04_j.sh): context for the roofline fractions, never part of the product path.
    python tools/bench_vendor_gemm.py [M] [iters]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cellvit_amd import _lib  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, N, K, act in (("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 0)):
        A = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
        W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5).half()
        b = torch.zeros(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        for fn, tag in ((lambda: lib.cv_op_linear(0, p(A), p(W), p(b), None, p(out), 0, M, N, K, act, None), "ours"),
                        (lambda: torch.matmul(A, W.t(), out=out), "vendor")):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            print(f"{name} M={M} N={N} K={K} {tag:6s}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:6.0f} TFLOP/s")
        del A, W, out


if __name__ == "__main__":
    main()
