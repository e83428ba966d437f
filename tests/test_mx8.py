"""CPU: OCP MX-fp8 host pieces of the fp8 engine — the library's quantiser (what cv_finalize applies to the encoder's
qkv / fc1 / fc2 weights) against an independent restatement built on torch's float8_e4m3fn cast, and the tiled scale
images (cellvit_amd/csrc/gemm.h) as permutations."""
import numpy as np
import pytest
import torch

from cellvit_amd import mx8


def _mx8_reference(x: np.ndarray):
    """OCP MX v1.0, e4m3 elements, block 32: shared exponent floor(log2(amax)) - emax(=8); elements saturate at +-448."""
    rows, K = x.shape
    xb = x.reshape(rows, K // 32, 32).astype(np.float32)
    amax = np.abs(xb).max(-1)
    ex = (amax.view(np.uint32) >> 23) & 0xff                   # biased exponent of amax (0 for zero / subnormal)
    sb = np.maximum(ex.astype(np.int32) - 8, 0).astype(np.uint8)
    inv = np.exp2(127.0 - sb.astype(np.float32)).astype(np.float32)
    q = np.clip(xb * inv[..., None], -448.0, 448.0).astype(np.float32)
    data = torch.from_numpy(q).to(torch.float8_e4m3fn).view(torch.uint8).numpy().reshape(rows, K)
    return data, sb


def test_e4m3_decode_table_matches_torch():
    b = np.arange(256, dtype=np.uint8)
    want = torch.from_numpy(b).view(torch.float8_e4m3fn).float().numpy()
    got = mx8.e4m3_to_f32(b)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    assert got[0x7e] == 448.0 and got[0x01] == 2.0 ** -9


@pytest.mark.parametrize("scale", [1.0, 1e-3, 300.0])
def test_host_quantiser_matches_the_ocp_restatement(scale):
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((256, 384)) * scale).astype(np.float32)
    x[3, :32] = 0.0                                              # an all-zero block
    x[7, 40] = 1e30                                              # a block dominated by one huge value
    x[9, 64:96] = 2.0 ** -140                                    # denormal inputs
    x[11, 100] = -448.0 * 2.0 ** 20
    data, sc = mx8.quantize(x, mx8.ROW_MAJOR)
    sc = sc.reshape(256, 12)
    ref_data, ref_sc = _mx8_reference(x)
    assert np.array_equal(sc, ref_sc)
    assert np.array_equal(data, ref_data), f"{(data != ref_data).sum()} element bytes differ"
    deq = mx8.dequantize(data, sc)
    blk_max = np.abs(x.reshape(256, 12, 32)).max(-1, keepdims=True).repeat(32, -1).reshape(256, 384)
    # relative error <= 2^-4 of the element (3 mantissa bits; 2^-3 for the block maximum when it lands in (448, 512) * 2^shared
    # and saturates — OCP MX's known clipping) down to the block's subnormal step 2^-9 * 2^(shared exp)
    err = np.abs(deq - x)
    fin = (np.abs(x) < 1e20) & (blk_max > 2.0 ** -100)           # (E8M0 bottoms out at 2^-127: tiny blocks flush to zero)
    assert (err[fin] <= np.maximum(np.abs(x) * 2.0 ** -3, blk_max * 2.0 ** -17 + 1e-45)[fin]).all()
    inner = np.abs(x) * 2.0 ** 8 < blk_max * 1.7                 # elements that cannot saturate
    assert (err[inner & fin] <= np.maximum(np.abs(x) * 2.0 ** -4, blk_max * 2.0 ** -17 + 1e-45)[inner & fin]).all()


def test_tiled_scale_images_are_permutations_of_the_row_major_view():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((512, 256)).astype(np.float32) * np.exp2(rng.integers(-6, 6, (512, 1))).astype(np.float32)
    d2, s2 = mx8.quantize(x, mx8.ROW_MAJOR)
    for layout, w_side in ((mx8.A_SIDE, False), (mx8.W_SIDE, True)):
        d, s = mx8.quantize(x, layout)
        assert np.array_equal(d, d2)
        idx = mx8.scale_index(512, 256, w_side)
        assert sorted(idx.reshape(-1).tolist()) == list(range(512 * 8))          # a bijection
        assert np.array_equal(mx8.untile_scales(s, 512, 256, w_side), s2.reshape(512, 8))
        assert np.array_equal(mx8.tile_scales(s2.reshape(512, 8), w_side), s)
    # one ds_read_b32 of lane (g, li) must hold the four fragment scales of a sub-tile: A side rows mi*16 + li (mi = byte)
    ia = mx8.scale_index(256, 128, False)
    for wr in range(2):
        for mh in range(2):
            for g in range(4):
                for li in range(16):
                    base = ((wr * 2 + mh) * 64 + g * 16 + li) * 4
                    assert [int(ia[wr * 128 + mh * 64 + mi * 16 + li, g]) for mi in range(4)] == [base + mi for mi in range(4)]
    # W side: lane li = q*4 + r of fragment nj reads source row wc*64 + q*16 + nj*4 + r
    iw = mx8.scale_index(256, 128, True)
    for wc in range(4):
        for g in range(4):
            for li in range(16):
                base = (wc * 64 + g * 16 + li) * 4 + 1024 * 0
                q, r = li >> 2, li & 3
                assert [int(iw[wc * 64 + q * 16 + nj * 4 + r, g]) for nj in range(4)] == [base + nj for nj in range(4)]
