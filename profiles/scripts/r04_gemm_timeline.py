"""Per-tile timeline of the 8-phase linear GEMM (ablation flavour, CVA_GEMM_DBG=32768): wave 0 / wave 4 of the first 8 workgroups stamp the 100-MHz wall clock
at  0 tile top | 1 first K tile landed (vmcnt) | 2 K loop done | 3 loads drained + wave groups re-aligned | 4 next tile's DMA issued, bias read | 5 epilogue done.
    CVA_LIB=abl CVA_GEMM_DBG=32768 CVA_GEMM_PHASE=0 python tools/experiments/r04_gemm_timeline.py M N K act"""
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
t = np.array(buf[:], dtype=np.int64).reshape(8, 2, 24, 8)[..., :6].astype(np.float64) * 0.01      # us
names = ["wait first K tile", "K loop", "drain + realign", "setup + DMA issue + bias", "epilogue", "-> next tile top"]
seg = np.stack([t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2], t[..., 4] - t[..., 3], t[..., 5] - t[..., 4]], -1)   # [wg, grp, tile, 5]
nxt = t[:, :, 1:, 0] - t[:, :, :-1, 5]
tile = t[:, :, 1:, 0] - t[:, :, :-1, 0]
ntl = (M // 256) * (N // 256) // 256   # tiles per workgroup (256 persistent workgroups); stamps exist for the first 24
sl = slice(3, min(ntl, 24) - 2)        # steady-state tiles
print(f"M={M} N={N} K={K} act={act}: tile period {tile[:, :, sl].mean():.2f} us (8 workgroups x 2 wave groups x {sl.stop - sl.start} tiles, mean / min / max per segment, us)")
for i, nm in enumerate(names[:5]):
    v = seg[:, :, sl, i]
    print(f"  {nm:28s} {v.mean():7.2f} {v.min():7.2f} {v.max():7.2f}")
v = nxt[:, :, sl]
print(f"  {names[5]:28s} {v.mean():7.2f} {v.min():7.2f} {v.max():7.2f}")
