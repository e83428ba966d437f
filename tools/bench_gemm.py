"""Micro-benchmark of the linear GEMM entry point (A/B of kernel variants via CVA_GEMM).
    CVA_GEMM=4 python tools/bench_gemm.py [M N K] [iters]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cellvit_amd import _lib  # noqa: E402

if os.environ.get("CVA_LIB") == "abl":      # the experiment flavour (python -m cellvit_amd.build --ablation): CVA_* switches honoured
    _lib.LIB_PATH = os.path.join(ROOT, "cellvit_amd", "libcellvit_amd_abl.so")


def main():
    M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (16384, 5120, 1280)
    iters = int(sys.argv[4]) if len(sys.argv) >= 5 else 20
    lib = _lib.load()
    act = int(os.environ.get("ACT", "0"))
    use_res = int(os.environ.get("RES", "0"))
    g = torch.Generator(device="cuda").manual_seed(0)
    A = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
    W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5).half()
    b = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if use_res else torch.float16)
    res = torch.randn(M, N, device="cuda") if use_res else None
    pr = C.c_void_p(res.data_ptr()) if use_res else None
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for _ in range(3):
        _lib.check(lib.cv_op_linear(0, p(A), p(W), p(b), pr, p(out), use_res, M, N, K, act, None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.cv_op_linear(0, p(A), p(W), p(b), pr, p(out), use_res, M, N, K, act, None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    # full-matrix check (in row blocks) + race screen: repeated launches must reproduce the first result bit for bit
    err = 0.0
    for r0 in range(0, M, 4096):
        ref = A[r0:r0 + 4096].float() @ W.float().t()
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        if use_res:
            ref = ref + res[r0:r0 + 4096]
        err = max(err, float((out[r0:r0 + 4096].float() - ref).abs().max()))
    first = out.clone()
    nrace = int(os.environ.get("RACE", "5"))
    bad = 0
    for _ in range(nrace):
        out.zero_()
        lib.cv_op_linear(0, p(A), p(W), p(b), pr, p(out), use_res, M, N, K, act, None)
        torch.cuda.synchronize()
        bad += int((out != first).sum().item())
    if bad:
        print(f"RACE: {bad} mismatching elements over {nrace} repeats")
    print(f"CVA_GEMM4={os.environ.get('CVA_GEMM4', '-')} ACT={act} RES={use_res} M={M} N={N} K={K}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s  maxerr {err:.3e}")


if __name__ == "__main__":
    main()
