"""GPU: every HIP kernel family, driven through the C ABI, against a plain PyTorch fp32 reference
of the same operator (CPU).  fp32 storage path: tight tolerance; fp16 path: fp16-rounding tolerance."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

F16, F32 = 0, 1
TOL = {F32: 2e-4, F16: 3e-2}


def _lib():
    from cellvit_amd import _lib as L
    return L, L.load()


def _dev(t, dtype):
    return t.to("cuda", dtype=torch.float16 if dtype == F16 else torch.float32).contiguous()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _rel_err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-6))


@pytest.mark.parametrize("dtype", [F32, F16])
@pytest.mark.parametrize("M,N,K,act,res", [(300, 200, 256, 1, True), (128, 128, 64, 0, False), (77, 19, 384, 0, False),
                                           (1000, 1152, 384, 2, True)])
def test_linear(dtype, M, N, K, act, res):
    L, lib = _lib()
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    R = torch.randn(M, N, generator=g) if res else None
    Ad, Wd = _dev(A, dtype), _dev(W, dtype)
    bd = b.cuda()
    Rd = R.cuda() if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    L.check(lib.cv_op_linear(dtype, _p(Ad), _p(Wd), _p(bd), _p(Rd), _p(out), 1, M, N, K, act, None))
    torch.cuda.synchronize()
    ref = F.linear(Ad.float().cpu(), Wd.float().cpu(), b)
    ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    if res:
        ref = ref + R
    err = _rel_err(out.cpu(), ref)
    assert err < (1e-5 if dtype == F32 else 1e-3), err   # operands identical (pre-rounded): only accumulation order
    # storage-typed output
    out2 = torch.empty(M, N, device="cuda", dtype=torch.float16 if dtype == F16 else torch.float32)
    L.check(lib.cv_op_linear(dtype, _p(Ad), _p(Wd), _p(bd), _p(Rd), _p(out2), 0, M, N, K, act, None))
    torch.cuda.synchronize()
    assert _rel_err(out2.float().cpu(), ref) < (1e-5 if dtype == F32 else 2e-3)


@pytest.mark.parametrize("dtype", [F32, F16])
@pytest.mark.parametrize("M,Cc", [(257, 384), (100, 1280), (64, 256), (5, 768)])
def test_layernorm_rows(dtype, M, Cc):
    L, lib = _lib()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(M, Cc, generator=g) * 3 + 0.5
    w = torch.rand(Cc, generator=g) + 0.5
    b = torch.randn(Cc, generator=g) * 0.1
    out = torch.empty(M, Cc, device="cuda", dtype=torch.float16 if dtype == F16 else torch.float32)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()   # keep the device tensors alive across the call
    L.check(lib.cv_op_layernorm(dtype, _p(xd), _p(wd), _p(bd), _p(out), 0, M, Cc, 1e-6, None))
    torch.cuda.synchronize()
    ref = F.layer_norm(x, (Cc,), w, b, 1e-6)
    assert _rel_err(out.float().cpu(), ref) < (1e-5 if dtype == F32 else 1e-3)


@pytest.mark.parametrize("M,Cc,alias", [(257, 384, False), (100, 1280, True), (4096, 1280, True), (5, 768, False)])
def test_layernorm_add_rows(M, Cc, alias):
    """residual add fused into the following LayerNorm (fp16 engine): x += delta; out = LN(x); out may alias delta."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, Cc, generator=g) * 3 + 0.5
    d = (torch.randn(M, Cc, generator=g) * 0.7).half()
    w = torch.rand(Cc, generator=g) + 0.5
    b = torch.randn(Cc, generator=g) * 0.1
    xd, dd, wd, bd = x.cuda(), d.cuda(), w.cuda(), b.cuda()
    out = dd if alias else torch.empty(M, Cc, device="cuda", dtype=torch.float16)
    L.check(lib.cv_op_layernorm_add(_p(xd), _p(dd), _p(wd), _p(bd), _p(out), M, Cc, 1e-6, None))
    torch.cuda.synchronize()
    xs = x + d.float()
    assert torch.equal(xd.cpu(), xs)                       # the fp32 stream is updated exactly
    ref = F.layer_norm(xs, (Cc,), w, b, 1e-6)
    assert _rel_err(out.float().cpu(), ref) < 1e-3


def _pack_conv(W):  # [Cout, Cin, 3, 3] -> [Cout, 9*Cin], k = tap*Cin + c
    Cout, Cin = W.shape[:2]
    return W.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()


@pytest.mark.parametrize("dtype", [F32, F16])
@pytest.mark.parametrize("B,H,W_,C1,C2,Cout,relu", [(2, 20, 24, 64, 64, 96, 1), (1, 33, 17, 128, 0, 64, 0),
                                                      (1, 16, 16, 312, 312, 312, 1)])
def test_conv3x3(dtype, B, H, W_, C1, C2, Cout, relu):
    _conv3x3_case(dtype, B, H, W_, C1, C2, Cout, relu, 1)


@pytest.mark.parametrize("B,H,W_,C1,C2,Cout,relu", [(2, 16, 16, 64, 64, 256, 1), (1, 32, 8, 128, 0, 512, 0), (3, 16, 32, 256, 256, 256, 1),
                                                      (1, 8, 256, 128, 0, 256, 1), (1, 2, 512, 64, 64, 256, 1)])
def test_conv3x3_implicit_gemm_shapes(B, H, W_, C1, C2, Cout, relu):
    """fp16 layers with Cout % 256 == 0 and power-of-two sides run as an implicit GEMM on the 8-phase kernel (tap-shifted
    LDS-DMA rows, zero padding by the descriptor range check): image corners / edges, two sources, rows wider than a tile."""
    _conv3x3_case(F16, B, H, W_, C1, C2, Cout, relu, 0)


@pytest.mark.parametrize("B,H,W_,C1,relu", [(2, 128, 128, 64, 1), (3, 256, 512, 64, 1), (1, 200, 300, 32, 1), (2, 144, 112, 32, 0), (1, 131, 260, 64, 0)])
def test_conv3x3_resident_filter_persistent(B, H, W_, C1, relu):
    """Cin 32 / 64 -> 64 channels, fp16 out: the persistent kernel with the filter resident in LDS (conv3x3_res_kernel): several tiles per
    workgroup (768 tiles on 256 CUs), ragged image sides (tiles cut by the right / bottom border), one and two channel chunks."""
    _conv3x3_case(F16, B, H, W_, C1, 0, 64, relu, 0)


def _conv3x3_case(dtype, B, H, W_, C1, C2, Cout, relu, out_f32):
    L, lib = _lib()
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(B, H, W_, C1, generator=g)
    x2 = torch.randn(B, H, W_, C2, generator=g) if C2 else None
    Wt = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt(9 * (C1 + C2))
    bias = torch.randn(Cout, generator=g) * 0.1
    x1d = _dev(x1, dtype)
    x2d = _dev(x2, dtype) if C2 else None
    Wd = _dev(_pack_conv(Wt), dtype)
    out = torch.full((B, H, W_, Cout), float("nan"), device="cuda", dtype=torch.float32 if out_f32 else torch.float16)
    biasd = bias.cuda()
    L.check(lib.cv_op_conv3x3(dtype, _p(x1d), C1, _p(x2d), C2, _p(Wd), _p(biasd), _p(out), out_f32, B, H, W_, Cout,
                              relu, None))
    torch.cuda.synchronize()
    xin = x1d.float().cpu() if not C2 else torch.cat([x1d.float().cpu(), x2d.float().cpu()], dim=-1)
    Wr = Wd.float().cpu().reshape(Cout, 3, 3, C1 + C2).permute(0, 3, 1, 2)
    ref = F.conv2d(xin.permute(0, 3, 1, 2), Wr, bias, padding=1)
    ref = F.relu(ref) if relu else ref
    err = _rel_err(out.float().cpu(), ref.permute(0, 2, 3, 1))
    assert err < (1e-5 if dtype == F32 else 1e-3 if out_f32 else 2e-3), err


@pytest.mark.parametrize("dtype", [F32, F16])
@pytest.mark.parametrize("B,H,W_,Cin,Cout", [(2, 8, 8, 128, 48), (1, 16, 12, 384, 312), (1, 5, 7, 64, 256)])
def test_convT2x2(dtype, B, H, W_, Cin, Cout):
    L, lib = _lib()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, H, W_, Cin, generator=g)
    Wt = torch.randn(Cin, Cout, 2, 2, generator=g) / math.sqrt(Cin)
    bias = torch.randn(Cout, generator=g) * 0.1
    xd = _dev(x, dtype)
    Wk = Wt.permute(2, 3, 1, 0).reshape(4 * Cout, Cin).contiguous()   # n = (dy*2+dx)*Cout + co
    Wd = _dev(Wk, dtype)
    b4 = bias.repeat(4).cuda()
    dt = torch.float16 if dtype == F16 else torch.float32
    out = torch.empty(B, 2 * H, 2 * W_, Cout, device="cuda", dtype=dt)
    L.check(lib.cv_op_convT2x2(dtype, _p(xd), _p(Wd), _p(b4), _p(out), B, H, W_, Cin, Cout, None))
    torch.cuda.synchronize()
    Wr = Wd.float().cpu().reshape(2, 2, Cout, Cin).permute(3, 2, 0, 1)
    ref = F.conv_transpose2d(xd.float().cpu().permute(0, 3, 1, 2), Wr, bias, stride=2).permute(0, 2, 3, 1)
    err = _rel_err(out.float().cpu(), ref)
    assert err < (1e-5 if dtype == F32 else 2e-3), err


@pytest.mark.parametrize("B,H,W_,Cin,Cs,Cout", [(2, 16, 16, 128, 0, 256), (1, 32, 16, 64, 0, 256), (1, 4, 64, 192, 0, 512), (3, 16, 16, 320, 0, 256),
                                                   (2, 16, 16, 128, 256, 256), (1, 8, 32, 64, 128, 256), (1, 2, 256, 192, 128, 512), (3, 16, 16, 128, 384, 256),
                                                   # the full-resolution stages (Cout 128 / 64: the halo kernel of deconv.hip — four output parities per workgroup,
                                                   # parity-split skip halo): the model's two shapes, no skip, several images, partial tiles in both directions
                                                   (2, 16, 16, 256, 128, 128), (1, 8, 16, 128, 64, 64), (2, 24, 48, 128, 64, 64), (1, 16, 16, 256, 0, 128),
                                                   (3, 5, 7, 64, 64, 64), (1, 9, 33, 64, 0, 64), (1, 32, 32, 192, 128, 128)])
def test_deconv_block_composed(B, H, W_, Cin, Cs, Cout):
    """ConvTranspose2d k2 s2 -> Conv2d 3x3 -> BatchNorm2d -> ReLU as one composed launch against the torch modules in fp32 on the same
    (fp16-rounded) inputs.  Cs == 0: Deconv2DBlock (models/segmentation/cell_segmentation/utils.py:46-86); Cs > 0: the convolution runs on
    torch.cat([skip, up-sampled], dim=1) (cellvit.py:236-242, 255-304).  Interior and all border cases of the 2H x 2W output, several
    images per launch, odd chunk counts, rows wider than a tile."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(11)
    Cup = Cout
    x = torch.randn(B, H, W_, Cin, generator=g)
    skip = torch.randn(B, 2 * H, 2 * W_, Cs, generator=g) if Cs else None
    wt = torch.randn(Cin, Cup, 2, 2, generator=g) / math.sqrt(Cin)
    bt = torch.randn(Cup, generator=g) * 0.3
    w3 = torch.randn(Cout, Cs + Cup, 3, 3, generator=g) / math.sqrt(9 * (Cs + Cup))
    b3 = torch.randn(Cout, generator=g) * 0.1
    bn_w = torch.rand(Cout, generator=g) + 0.5
    bn_b = torch.randn(Cout, generator=g) * 0.1
    bn_m = torch.randn(Cout, generator=g) * 0.1
    bn_v = torch.rand(Cout, generator=g) + 0.5
    xd = _dev(x, F16)
    sd = _dev(skip, F16) if Cs else None
    out = torch.full((B, 2 * H, 2 * W_, Cout), float("nan"), device="cuda", dtype=torch.float16)
    host = [t.contiguous() for t in (wt, bt, w3, b3, bn_w, bn_b, bn_m, bn_v)]       # host fp32 arrays
    L.check(lib.cv_op_deconv_block(*[C.c_void_p(t.data_ptr()) for t in host], _p(xd), _p(sd), _p(out), B, H, W_, Cin, Cup, Cs, Cout, None))
    torch.cuda.synchronize()
    up = F.conv_transpose2d(xd.float().cpu().permute(0, 3, 1, 2), wt, bt, stride=2)
    cat = torch.cat([sd.float().cpu().permute(0, 3, 1, 2), up], dim=1) if Cs else up
    y = F.conv2d(cat, w3, b3, padding=1)
    y = F.batch_norm(y, bn_m, bn_v, bn_w, bn_b, training=False, eps=1e-5)
    ref = F.relu(y).permute(0, 2, 3, 1)
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    err = _rel_err(got, ref)
    assert err < 2e-3, err
    # the border rows / columns on their own (a wrong bias case or tap mask would hide in the global norm)
    for sl in (got[:, 0] - ref[:, 0], got[:, -1] - ref[:, -1], got[:, :, 0] - ref[:, :, 0], got[:, :, -1] - ref[:, :, -1]):
        assert float(sl.abs().max()) < 2e-2 * float(ref.abs().max()), float(sl.abs().max())


def _attention_ref(x, Wqkv, bqkv, tab_h, tab_w, B, gh, gw, has_cls, heads, D, win):
    """torch fp32 reference: ViT attention (vits_histo.py:174-185) or SAM attention with window
    partition + decomposed rel-pos (image_encoder.py:181-188, 235-257, 354-392), without out-proj."""
    hd = D // heads
    ntok = gh * gw + has_cls
    x = x.reshape(B, ntok, D)
    if tab_h is None:
        qkv = F.linear(x, Wqkv, bqkv).reshape(B, ntok, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
        return (a @ v).transpose(1, 2).reshape(B * ntok, D)

    def attn(xw):  # [B', h, w, D]
        Bp, h, w, _ = xw.shape
        qkv = F.linear(xw, Wqkv, bqkv).reshape(Bp, h * w, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.reshape(3, Bp * heads, h * w, hd).unbind(0)
        a = (q * hd ** -0.5) @ k.transpose(-2, -1)
        idx_h = (torch.arange(h)[:, None] - torch.arange(h)[None, :]) + (h - 1)
        idx_w = (torch.arange(w)[:, None] - torch.arange(w)[None, :]) + (w - 1)
        Rh, Rw = tab_h[idx_h], tab_w[idx_w]
        rq = q.reshape(Bp * heads, h, w, hd)
        rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
        rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
        a = (a.view(-1, h, w, h, w) + rel_h[..., None] + rel_w[..., None, :]).view(-1, h * w, h * w).softmax(-1)
        return (a @ v).view(Bp, heads, h, w, hd).permute(0, 2, 3, 1, 4).reshape(Bp, h, w, D)

    x = x.reshape(B, gh, gw, D)
    if win > 0:
        ph, pw = (win - gh % win) % win, (win - gw % win) % win
        xp = F.pad(x, (0, 0, 0, pw, 0, ph))
        Hp, Wp = gh + ph, gw + pw
        xw = xp.view(B, Hp // win, win, Wp // win, win, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, win, win, D)
        y = attn(xw).view(B, Hp // win, Wp // win, win, win, D).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, D)
        y = y[:, :gh, :gw]
    else:
        y = attn(x)
    return y.reshape(B * gh * gw, D)


@pytest.mark.parametrize("dtype", [F32, F16])
@pytest.mark.parametrize("B,gh,gw,has_cls,heads,D,win,rel", [
    (2, 5, 7, 1, 6, 384, 0, False),      # ViT-S: cls token, ragged key count (36 keys)
    (1, 16, 16, 1, 6, 384, 0, False),    # ViT-S @256: 257 keys -> 5 key tiles
    (1, 8, 8, 0, 12, 768, 0, True),      # SAM-B global, hd 64, rel-pos
    (2, 16, 16, 0, 12, 768, 14, True),   # SAM-B window: 16 -> pad 28, 4 windows, padded keys live
    (1, 16, 16, 0, 16, 1280, 0, True),   # SAM-H global, hd 80
    (1, 16, 16, 0, 16, 1280, 14, True),  # SAM-H window, hd 80
    (1, 14, 14, 0, 16, 1280, 14, True),  # exact window fit (no padding)
    (1, 32, 32, 0, 4, 320, 0, True),     # 512-px tile global block: KH+KW = 64 -> two one-hot bias k-steps
    (1, 64, 64, 0, 4, 320, 0, True),     # 1024-px tile global block: key-tile-aligned bias path, 4096 keys
    (2, 64, 64, 0, 4, 256, 14, True),    # 1024-px tile window blocks: 64 -> pad 70, 25 windows
    (2, 64, 64, 0, 4, 320, 0, True),     # the same global block, two images
    (1, 32, 64, 0, 4, 320, 0, True),     # key-tile-aligned bias with KH < 64 (2048 keys)
    (1, 33, 64, 0, 4, 320, 0, True),     # KH not a multiple of 4 (element-wise kh stores of the LDS-DMA kernel's prologue), 2112 queries: the last workgroup half idle
    (1, 64, 64, 1, 6, 384, 0, False),    # ViT-S @1024: 4097 keys (cls token), last key tile holds ONE key, last query block one query
    (1, 20, 20, 0, 4, 320, 0, False),    # hd 80 without bias, 400 keys: ragged last tile (one key block of four)
    (1, 24, 24, 1, 6, 384, 0, False),    # 577 keys: three query blocks of 256, the last with 65 queries
])
def test_attention(dtype, B, gh, gw, has_cls, heads, D, win, rel):
    L, lib = _lib()
    g = torch.Generator().manual_seed(5)
    hd = D // heads
    ntok = gh * gw + has_cls
    x = torch.randn(B * ntok, D, generator=g)
    Wqkv = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bqkv = torch.randn(3 * D, generator=g) * 0.3
    KH, KW = (win, win) if win else (gh, gw)
    tab_h = torch.randn(2 * KH - 1, hd, generator=g) * 0.2 if rel else None
    tab_w = torch.randn(2 * KW - 1, hd, generator=g) * 0.2 if rel else None
    xd, Wd = _dev(x, dtype), _dev(Wqkv, dtype)
    out = torch.zeros(B * ntok, D, device="cuda", dtype=torch.float16 if dtype == F16 else torch.float32)
    bd = bqkv.cuda()
    thd, twd = (tab_h.cuda(), tab_w.cuda()) if rel else (None, None)
    L.check(lib.cv_op_attention(dtype, _p(xd), _p(Wd), _p(bd), _p(thd), _p(twd), _p(out), B, gh, gw, has_cls, heads,
                                D, win, None))
    torch.cuda.synchronize()
    ref = _attention_ref(xd.float().cpu(), Wd.float().cpu(), bqkv, tab_h, tab_w, B, gh, gw, has_cls, heads, D, win)
    err = _rel_err(out.float().cpu(), ref)
    assert err < (2e-5 if dtype == F32 else 1e-2), err
