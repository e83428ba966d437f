"""Diagnostic for the MX-fp8 contraction (GPU box): crafted operands that isolate data path / scale path / indexing."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cellvit_amd import _lib, mx8  # noqa: E402

lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731


def run(a8, sa_rm, w8, sw_rm, M, N, K):
    sa = mx8.tile_scales(sa_rm, False)
    sw = mx8.tile_scales(sw_rm, True)
    ta, tsa, tw, tsw = dev(a8), dev(sa), dev(w8), dev(sw)
    out = torch.empty((M, N), device="cuda", dtype=torch.float32)
    _lib.check(lib.cv_op_linear_mx8(p(ta), p(tsa), None, p(tw), p(tsw), None, None, p(out), 1, None, M, N, K, 0, None))
    torch.cuda.synchronize()
    ref = mx8.dequantize(a8, sa_rm) @ mx8.dequantize(w8, sw_rm).T
    got = out.cpu().numpy().astype(np.float64)
    return got, ref


def report(name, got, ref):
    bad = np.abs(got - ref) > 1e-3 * (np.abs(ref) + 1)
    print(f"{name}: {int(bad.sum())} / {bad.size} wrong; got[0,:4]={got[0, :4]} ref[0,:4]={ref[0, :4]}")
    if bad.any():
        r, c = np.nonzero(bad)
        print("   wrong rows (first 12 distinct):", sorted(set(r.tolist()))[:12], " cols:", sorted(set(c.tolist()))[:12])
        print("   ratio got/ref at a few wrong spots:", [(int(i), int(j), float(got[i, j] / ref[i, j]) if ref[i, j] else None) for i, j in list(zip(r, c))[:6]])


def main():
    for (M, N, K) in [(256, 256, 256), (256, 256, 1280), (512, 512, 256)]:
        print(f"=== M={M} N={N} K={K}")
        one = np.full((M, K), 0x38, np.uint8)
        wone = np.full((N, K), 0x38, np.uint8)
        s1a = np.full((M, K // 32), 127, np.uint8)
        s1w = np.full((N, K // 32), 127, np.uint8)
        report("T1 ones, unit scales", *run(one, s1a, wone, s1w, M, N, K))
        rng = np.random.default_rng(0)
        small = np.array([0x00, 0x38, 0x40, 0x44, 0xb8, 0xc0], np.uint8)            # 0, 1, 2, 3, -1, -2
        a = small[rng.integers(0, 6, (M, K))]
        w = small[rng.integers(0, 6, (N, K))]
        report("T2 random A data, W ones", *run(a, s1a, wone, s1w, M, N, K))
        report("T3 A ones, random W data", *run(one, s1a, w, s1w, M, N, K))
        report("T4 random A and W data", *run(a, s1a, w, s1w, M, N, K))
        sa = (127 + (np.arange(M)[:, None] % 3) + 0 * np.arange(K // 32)[None, :]).astype(np.uint8)
        report("T5 ones, A scale by row", *run(one, sa, wone, s1w, M, N, K))
        sa = (127 + 0 * np.arange(M)[:, None] + (np.arange(K // 32)[None, :] % 4)).astype(np.uint8)
        report("T6 ones, A scale by k-block", *run(one, sa, wone, s1w, M, N, K))
        sw = (127 + (np.arange(N)[:, None] % 3) + 0 * np.arange(K // 32)[None, :]).astype(np.uint8)
        report("T7 ones, W scale by row", *run(one, s1a, wone, sw, M, N, K))
        sw = (127 + 0 * np.arange(N)[:, None] + (np.arange(K // 32)[None, :] % 4)).astype(np.uint8)
        report("T8 ones, W scale by k-block", *run(one, s1a, wone, sw, M, N, K))
        sa = rng.integers(120, 132, (M, K // 32)).astype(np.uint8)
        sw = rng.integers(120, 132, (N, K // 32)).astype(np.uint8)
        report("T9 random everything", *run(a, sa, w, sw, M, N, K))


if __name__ == "__main__":
    main()


def main2():
    M = N = 256
    K = 512
    one = np.full((M, K), 0x38, np.uint8)
    wone = np.full((N, K), 0x38, np.uint8)
    s1a = np.full((M, K // 32), 127, np.uint8)
    s1w = np.full((N, K // 32), 127, np.uint8)
    # unique scale per row: which row's scale does row r get?
    sa = (100 + (np.arange(M)[:, None] % 50) + 0 * np.arange(K // 32)[None, :]).astype(np.uint8)
    got, ref = run(one, sa, wone, s1w, M, N, K)
    e = np.round(np.log2(got[:, 0] / K)).astype(int) + 127
    print("A: scale exponent seen by row r (expected 100 + r % 50); mismatching rows:", [(r, int(e[r])) for r in range(M) if e[r] != 100 + r % 50][:16])
    sw = (100 + (np.arange(N)[:, None] % 50) + 0 * np.arange(K // 32)[None, :]).astype(np.uint8)
    got, ref = run(one, s1a, wone, sw, M, N, K)
    e = np.round(np.log2(got[0, :] / K)).astype(int) + 127
    print("W: scale exponent seen by col n; mismatching:", [(n, int(e[n])) for n in range(N) if e[n] != 100 + n % 50][:16])
    # k mapping: A data nonzero only in k-block q, A scale distinct per k-block
    sa = (110 + 0 * np.arange(M)[:, None] + np.arange(K // 32)[None, :]).astype(np.uint8)
    for q in range(K // 32):
        a = np.zeros((M, K), np.uint8)
        a[:, q * 32:(q + 1) * 32] = 0x38
        got, ref = run(a, sa, wone, s1w, M, N, K)
        e = np.round(np.log2(got[:, 0] / 32)).astype(int) + 127
        print(f"A data in k-block {q:2d}: scale exponent applied (expected {110 + q}): rows 0,17,70,200 ->", e[[0, 17, 70, 200]])
    sw = (110 + 0 * np.arange(N)[:, None] + np.arange(K // 32)[None, :]).astype(np.uint8)
    for q in range(K // 32):
        w = np.zeros((N, K), np.uint8)
        w[:, q * 32:(q + 1) * 32] = 0x38
        got, ref = run(one, s1a, w, sw, M, N, K)
        e = np.round(np.log2(got[0, :] / 32)).astype(int) + 127
        print(f"W data in k-block {q:2d}: scale exponent applied (expected {110 + q}): cols 0,17,70,200 ->", e[[0, 17, 70, 200]])


if __name__ == "__main__":
    main2()
