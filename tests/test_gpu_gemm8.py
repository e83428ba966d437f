"""GPU: shapes that route to the 8-phase 256x256 fp16 contraction kernel (gemm8.hip: M, N multiples of 256,
K multiple of 128, >= 192 tiles), through the C ABI, against a PyTorch fp32 reference of the same operator."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import _attention_ref, _dev, _lib, _p, _rel_err, F16

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,act,res", [(4096, 4096, 256, 1, True), (8192, 2048, 384, 0, False),
                                           (4096, 3072, 1280, 2, True), (16384, 1280, 128, 1, False)])
def test_linear_8phase(M, N, K, act, res):
    L, lib = _lib()
    g = torch.Generator().manual_seed(11)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    R = torch.randn(M, N, generator=g) if res else None
    Ad, Wd = _dev(A, F16), _dev(W, F16)
    bd = b.cuda()
    Rd = R.cuda() if res else None
    ref = F.linear(Ad.float(), Wd.float(), bd)
    ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    if res:
        ref = ref + Rd
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    L.check(lib.cv_op_linear(F16, _p(Ad), _p(Wd), _p(bd), _p(Rd), _p(out), 1, M, N, K, act, None))
    torch.cuda.synchronize()
    assert _rel_err(out, ref) < 1e-3
    out2 = torch.empty(M, N, device="cuda", dtype=torch.float16)
    first = None
    for _ in range(3):                      # race screen: repeated launches reproduce bit for bit
        out2.zero_()
        L.check(lib.cv_op_linear(F16, _p(Ad), _p(Wd), _p(bd), _p(Rd), _p(out2), 0, M, N, K, act, None))
        torch.cuda.synchronize()
        if first is None:
            first = out2.clone()
        assert torch.equal(out2, first)
    assert _rel_err(out2.float(), ref) < 2e-3


@pytest.mark.parametrize("B,gh,gw,heads,D,win", [(2, 64, 64, 16, 1280, 14), (1, 64, 64, 16, 1280, 14), (1, 64, 64, 4, 256, 0)])
def test_attention_qkv_8phase(B, gh, gw, heads, D, win):
    """qkv projection of a 1024-px tile: q/k columns through the transposed-accumulator direct epilogue,
    v columns through the staged V^T epilogue; checked end to end through the attention layer."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(5)
    hd = D // heads
    ntok = gh * gw
    x = torch.randn(B * ntok, D, generator=g)
    Wqkv = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bqkv = torch.randn(3 * D, generator=g) * 0.3
    KH, KW = (win, win) if win else (gh, gw)
    tab_h = torch.randn(2 * KH - 1, hd, generator=g) * 0.2
    tab_w = torch.randn(2 * KW - 1, hd, generator=g) * 0.2
    xd, Wd = _dev(x, F16), _dev(Wqkv, F16)
    out = torch.zeros(B * ntok, D, device="cuda", dtype=torch.float16)
    bd, thd, twd = bqkv.cuda(), tab_h.cuda(), tab_w.cuda()      # keep the device tensors alive across the call
    L.check(lib.cv_op_attention(F16, _p(xd), _p(Wd), _p(bd), _p(thd), _p(twd), _p(out),
                                B, gh, gw, 0, heads, D, win, None))
    torch.cuda.synchronize()
    ref = _attention_ref(xd.float().cpu(), Wd.float().cpu(), bqkv, tab_h, tab_w, B, gh, gw, 0, heads, D, win)
    assert _rel_err(out.float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("B,gh,gw,has_cls,heads,D,win,rel", [
    (2, 14, 14, 0, 6, 384, 0, False),     # 196 keys, hd 64, no bias   -> attnw_kernel<64, 0>
    (2, 14, 14, 1, 6, 384, 0, False),     # 197 keys (cls token), ragged last key block
    (1, 28, 28, 0, 12, 768, 14, True),    # SAM-B windows, hd 64, rel-pos -> attnw_kernel<64, 1>
    (1, 14, 14, 0, 16, 1280, 0, False),   # hd 80, no bias              -> attnw_kernel<80, 0>
    (1, 30, 30, 0, 16, 1280, 14, True),   # SAM-H windows with padded edge windows (30 -> 42) -> attnwp_kernel<80, 1>
    (2, 28, 28, 0, 12, 768, 14, True),    # 96 (window, head) items of hd 64 -> persistent attnwp_kernel<64, 1>
])
def test_attention_short_sequences(B, gh, gw, has_cls, heads, D, win, rel):
    """fp16 sequences of <= 208 keys run the single-pass window kernel (attention_win.hip)."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(7)
    hd = D // heads
    ntok = gh * gw + has_cls
    x = torch.randn(B * ntok, D, generator=g)
    Wqkv = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bqkv = torch.randn(3 * D, generator=g) * 0.3
    KH, KW = (win, win) if win else (gh, gw)
    tab_h = torch.randn(2 * KH - 1, hd, generator=g) * 0.2 if rel else None
    tab_w = torch.randn(2 * KW - 1, hd, generator=g) * 0.2 if rel else None
    xd, Wd = _dev(x, F16), _dev(Wqkv, F16)
    out = torch.zeros(B * ntok, D, device="cuda", dtype=torch.float16)
    bd = bqkv.cuda()
    thd, twd = (tab_h.cuda(), tab_w.cuda()) if rel else (None, None)
    L.check(lib.cv_op_attention(F16, _p(xd), _p(Wd), _p(bd), _p(thd), _p(twd), _p(out), B, gh, gw, has_cls, heads,
                                D, win, None))
    torch.cuda.synchronize()
    ref = _attention_ref(xd.float().cpu(), Wd.float().cpu(), bqkv, tab_h, tab_w, B, gh, gw, has_cls, heads, D, win)
    assert _rel_err(out.float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("B,H,W_,Cin,Cout", [(2, 64, 64, 1280, 512), (4, 128, 128, 512, 256), (8, 256, 256, 128, 64)])
def test_convT2x2_8phase(B, H, W_, Cin, Cout):
    """ConvTranspose2d k2 s2 at decoder sizes: the GEMM qualifies for the 8-phase kernel and stores the pixel-shuffled
    output as direct 32-byte runs (transposed accumulators, epilogue8_direct OUT_CONVT)."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, H, W_, Cin, generator=g)
    Wt = torch.randn(Cin, Cout, 2, 2, generator=g) / math.sqrt(Cin)
    bias = torch.randn(Cout, generator=g) * 0.1
    xd = _dev(x, F16)
    Wk = Wt.permute(2, 3, 1, 0).reshape(4 * Cout, Cin).contiguous()   # n = (dy*2+dx)*Cout + co
    Wd = _dev(Wk, F16)
    b4 = bias.repeat(4).cuda()
    out = torch.empty(B, 2 * H, 2 * W_, Cout, device="cuda", dtype=torch.float16)
    L.check(lib.cv_op_convT2x2(F16, _p(xd), _p(Wd), _p(b4), _p(out), B, H, W_, Cin, Cout, None))
    torch.cuda.synchronize()
    Wr = Wd.float().reshape(2, 2, Cout, Cin).permute(3, 2, 0, 1)
    ref = F.conv_transpose2d(xd.float().permute(0, 3, 1, 2), Wr, bias.cuda(), stride=2).permute(0, 2, 3, 1)
    assert _rel_err(out.float(), ref) < 2e-3


def test_window_attention_persistent_many_items_and_race_screen():
    """8 tiles of 64x64 tokens = 3200 (window, head) items on 256 persistent workgroups (12-13 items each, both K images
    and the register prefetch in steady state): checked against the reference, and repeated launches must reproduce the
    first result bit for bit (the kernel hand-counts vmcnt / lgkmcnt)."""
    L, lib = _lib()
    B, gh, gw, heads, D, win = 8, 64, 64, 16, 1280, 14
    g = torch.Generator().manual_seed(11)
    hd = D // heads
    ntok = gh * gw
    x = torch.randn(B * ntok, D, generator=g)
    Wqkv = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bqkv = torch.randn(3 * D, generator=g) * 0.3
    tab_h = torch.randn(2 * win - 1, hd, generator=g) * 0.2
    tab_w = torch.randn(2 * win - 1, hd, generator=g) * 0.2
    xd, Wd = _dev(x, F16), _dev(Wqkv, F16)
    bd, thd, twd = bqkv.cuda(), tab_h.cuda(), tab_w.cuda()
    outs = []
    for _ in range(4):
        out = torch.zeros(B * ntok, D, device="cuda", dtype=torch.float16)
        L.check(lib.cv_op_attention(F16, _p(xd), _p(Wd), _p(bd), _p(thd), _p(twd), _p(out), B, gh, gw, 0, heads, D, win, None))
        torch.cuda.synchronize()
        outs.append(out)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # reference on two of the eight tiles (the CPU reference is O(tokens^2) per window only, but 8 tiles take a while)
    for b in (0, 7):
        sl = slice(b * ntok, (b + 1) * ntok)
        ref = _attention_ref(xd[sl].float().cpu(), Wd.float().cpu(), bqkv, tab_h, tab_w, 1, gh, gw, 0, heads, D, win)
        assert _rel_err(outs[0][sl].float().cpu(), ref) < 1e-2


def test_linear_8phase_operand_beyond_2_gib():
    """A of 2.7 GB (fc2 of a 64-tile batch: M = 64 * 4096 rows of 5120 halves): lane offsets are relative to the tile's
    first row, so the 8-phase kernel keeps serving batches past the old 2 GiB operand limit.  Row blocks from both ends
    and the middle of the operand against torch."""
    L, lib = _lib()
    M, N, K = 64 * 4096, 256, 5120
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.empty((M, K), device="cuda", dtype=torch.float16)
    for r0 in range(0, M, 32768):                              # fill in slabs (no fp32 temporary of the whole operand)
        A[r0:r0 + 32768] = torch.randn((32768, K), device="cuda", generator=g).half()
    W = (torch.randn((N, K), device="cuda", generator=g) / math.sqrt(K)).half()
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    assert A.numel() * 2 > (1 << 31)
    out = torch.empty((M, N), device="cuda", dtype=torch.float16)
    L.check(lib.cv_op_linear(F16, _p(A), _p(W), _p(b), None, _p(out), 0, M, N, K, 0, None))
    torch.cuda.synchronize()
    for r0 in (0, 100000, 209715 // 256 * 256, M - 4096):
        ref = F.linear(A[r0:r0 + 4096].float(), W.float(), b)
        assert _rel_err(out[r0:r0 + 4096].float(), ref) < 2e-3, r0
