"""Same-call comparison of the hot kernels with the vendor libraries torch dispatches to on ROCm (hipBLASLt / rocBLAS GEMM,
MIOpen convolution, the SDPA kernel) on the shapes of the SAM-H step.  Context for the roofline fractions only: the
vendor paths are never part of the product path.
    python tools/bench_vendor.py [tiles]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from cellvit_amd import _lib  # noqa: E402


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    M = tiles * 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    print(f"# linear layers, M = {M} (fp16 in, fp16 out, bias)")
    for name, N, K in (("qkv", 3840, 1280), ("proj", 1280, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120)):
        A = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
        W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5).half()
        b = torch.zeros(N, device="cuda")
        bh = b.half()
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        ours = timed(lambda: lib.cv_op_linear(0, p(A), p(W), p(b), None, p(out), 0, M, N, K, 0, None))
        vend = timed(lambda: F.linear(A, W, bh))
        vend_nb = timed(lambda: torch.matmul(A, W.t()))
        fl = 2.0 * M * N * K
        print(f"{name:5s} N={N} K={K}: ours {ours * 1e3:8.1f} us {fl / ours / 1e9:6.0f} TF | F.linear {vend * 1e3:8.1f} us {fl / vend / 1e9:6.0f} TF"
              f" | matmul {vend_nb * 1e3:8.1f} us {fl / vend_nb / 1e9:6.0f} TF")
        del A, W, out
    print("# 3x3 convolutions (NHWC fp16, bias; ours fuses BN + ReLU, the vendor call is the bare convolution)")
    bt = min(tiles, 8)
    for name, cin, cout, hw in (("deep 1024->512 @128", 1024, 512, 128), ("512->256 @256", 512, 256, 256), ("256->128 @512", 256, 128, 512),
                                ("64->64 @1024", 64, 64, 1024)):
        x = torch.randn(bt, cin, hw, hw, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, device="cuda") / (9 * cin) ** 0.5).half().contiguous(memory_format=torch.channels_last)
        bias = torch.zeros(cout, device="cuda", dtype=torch.float16)
        vend = timed(lambda: F.conv2d(x, w, bias, padding=1), iters=5, warm=2)
        fl = 2.0 * bt * hw * hw * cout * cin * 9
        print(f"{name:22s} B={bt}: MIOpen {vend * 1e3:8.1f} us {fl / vend / 1e9:6.0f} TF")
        del x, w
    print("# global attention core: 16 heads x 4096 tokens x 80 (no rel-pos in the vendor call)")
    q = torch.randn(bt, 16, 4096, 80, device="cuda", dtype=torch.float16)
    k, v = torch.randn_like(q), torch.randn_like(q)
    vend = timed(lambda: F.scaled_dot_product_attention(q, k, v), iters=5, warm=2)
    fl = 4.0 * bt * 16 * 4096 * 4096 * 80
    print(f"sdpa B={bt}: {vend * 1e3:8.1f} us {fl / vend / 1e9:6.0f} TF")


if __name__ == "__main__":
    main()
