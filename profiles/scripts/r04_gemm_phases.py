"""Shader-clock stamps at the top of each of the eight phases of the linear GEMM's K loop, third iteration of every tile (ablation flavour, CVA_GEMM_DBG=294912 =
32768 + 262144): how many core cycles each phase of the schedule takes (16 MFMAs = 256 matrix-pipe cycles per wave and phase; the two wave groups alternate).
    CVA_LIB=abl CVA_GEMM_DBG=294912 CVA_GEMM_PHASE=0 python tools/experiments/r04_gemm_phases.py M N K act"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cellvit_amd import _lib
M, N, K, act = (int(a) for a in sys.argv[1:5])
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5).half()
b = torch.zeros(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.float16)
p = lambda t: C.c_void_p(t.data_ptr())
for _ in range(3):
    _lib.check(lib.cv_op_linear(0, p(A), p(W), p(b), None, p(out), 0, M, N, K, act, None))
torch.cuda.synchronize()
n = 8 * 2 * 24 * 8
buf = (C.c_longlong * n)()
assert lib.cv_dbg_gemm_stamps(buf, n) == 0
t = np.array(buf[:], dtype=np.int64).reshape(8, 2, 24, 8).astype(np.float64)
ntl = (M // 256) * (N // 256) // 256
sl = slice(3, min(ntl, 24) - 2)
d = t[:, :, sl, 1:] - t[:, :, sl, :-1]                 # phases 1 .. 7 (the eighth ends at the next iteration's first stamp, not taken)
what = ["1", "2", "3", "4", "5", "6", "7"]      # (what each phase reads / stages: the table at the K loop in gemm8.hip)
print(f"M={M} N={N} K={K} act={act}: core cycles per phase (top of phase i -> top of phase i+1), mean / min / max over 8 workgroups x 2 wave groups x {sl.stop - sl.start} tiles")
for i, w in enumerate(what):
    v = d[..., i]
    print(f"  phase {w:34s} {v.mean():7.0f} {v.min():7.0f} {v.max():7.0f}")
print(f"  phases 1-7 together {d.sum(-1).mean():7.0f} cycles; two wave groups alternate, so one phase of one group ideally lasts 2 x 256 matrix-pipe cycles")
