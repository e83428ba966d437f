"""SHA-256 of the outputs of the implicit-GEMM convolution and of the composed ConvTranspose o Conv launch at the decoder's production shapes (several tiles per
workgroup, long K loops), on seeded inputs: two builds of the library whose kernels differ only in WHEN / BY WHICH WAVE the operand tiles are staged must print the
same digests (same MFMAs in the same order).   CVA_LIB=<lib>.so python tools/experiments/r04_conv_identity.py"""
import ctypes as C, hashlib, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cellvit_amd import _lib
lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def sha(t): return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]
B = 4
for H, C1, C2, Co in [(128, 512, 512, 512), (128, 512, 0, 512), (256, 256, 256, 256), (256, 256, 0, 256)]:
    g = torch.Generator(device="cuda").manual_seed(H + C2)
    s1 = torch.randn(B, H, H, C1, device="cuda", generator=g).half()
    s2 = torch.randn(B, H, H, C2, device="cuda", generator=g).half() if C2 else None
    W = (torch.randn(Co, 9 * (C1 + C2), device="cuda", generator=g) / (9 * (C1 + C2)) ** 0.5).half()
    b = torch.randn(Co, device="cuda", generator=g) * 0.1
    out = torch.empty(B, H, H, Co, device="cuda", dtype=torch.float16)
    _lib.check(lib.cv_op_conv3x3(0, p(s1), C1, p(s2), C2, p(W), p(b), p(out), 0, B, H, H, Co, 1, None))
    torch.cuda.synchronize()
    print(f"conv3x3 H={H} C={C1}+{C2}->{Co}: {sha(out)}  finite={bool(torch.isfinite(out).all())}")
for H, Cin, Cs, Cout in [(64, 1280, 512, 512), (128, 512, 256, 256), (64, 1280, 0, 512)]:
    g = torch.Generator().manual_seed(H + Cs)
    x = torch.randn(B, H, H, Cin, generator=g)
    skip = torch.randn(B, 2 * H, 2 * H, Cs, generator=g) if Cs else None
    wt = torch.randn(Cin, Cout, 2, 2, generator=g) / math.sqrt(Cin)
    bt = torch.randn(Cout, generator=g) * 0.3
    w3 = torch.randn(Cout, Cs + Cout, 3, 3, generator=g) / math.sqrt(9 * (Cs + Cout))
    b3 = torch.randn(Cout, generator=g) * 0.1
    bn = [torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1, torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5]
    xd = x.cuda().half(); sd = skip.cuda().half() if Cs else None
    out = torch.full((B, 2 * H, 2 * H, Cout), float("nan"), device="cuda", dtype=torch.float16)
    host = [t.contiguous() for t in (wt, bt, w3, b3, *bn)]
    _lib.check(lib.cv_op_deconv_block(*[C.c_void_p(t.data_ptr()) for t in host], p(xd), p(sd), p(out), B, H, H, Cin, Cout, Cs, Cout, None))
    torch.cuda.synchronize()
    print(f"deconv_block H={H} Cin={Cin} skip={Cs} -> {Cout}: {sha(out)}  finite={bool(torch.isfinite(out).all())}")
