"""Names of the kernels torch (hipBLASLt / rocBLAS) launches for the four linear shapes: they carry the vendor's macro-tile /
prefetch configuration.  (rocprofv3 around a hipBLASLt process did not finish in 10 minutes on the pool; torch.profiler does.)"""
import torch
from torch.profiler import ProfilerActivity, profile
M = 131072
mats = []
for N, K in ((3840, 1280), (1280, 1280), (5120, 1280), (1280, 5120)):
    A = torch.randn(M, K, device="cuda", dtype=torch.float16)
    W = torch.randn(N, K, device="cuda", dtype=torch.float16)
    torch.matmul(A, W.t())
    mats.append((A, W))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for A, W in mats:
        for _ in range(3):
            torch.matmul(A, W.t())
    torch.cuda.synchronize()
for e in prof.key_averages():
    print(f"{e.count:3d} x {e.device_time_total / max(e.count, 1):9.1f} us  {e.key}")
