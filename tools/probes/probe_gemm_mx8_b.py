"""Which scale (k-block index of the lane's scale byte) multiplies each 16-byte PIECE of a 128-byte K tile row."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from probe_gemm_mx8 import run  # noqa: E402

M = N = 256
K = 256
wone = np.full((N, K), 0x38, np.uint8)
one = np.full((M, K), 0x38, np.uint8)
s1 = np.full((M, K // 32), 127, np.uint8)
sa = (100 + 0 * np.arange(M)[:, None] + 8 * (np.arange(K // 32)[None, :] % 4)).astype(np.uint8)     # k-block g -> 2^(8g - 27)
for piece in range(16):
    a = np.zeros((M, K), np.uint8)
    a[:, piece * 16:(piece + 1) * 16] = 0x38
    got, _ = run(a, sa, wone, s1, M, N, K)
    v = got[5, 7] / 16.0
    print(f"A piece {piece:2d} (k-block {piece // 2 % 4}, tile {piece // 8}): scaled by 2^{np.log2(v) + 127 - 100:.0f}/8 -> lane scale of k-block {(np.log2(v) + 27) / 8:.2f}")
for piece in range(16):
    w = np.zeros((N, K), np.uint8)
    w[:, piece * 16:(piece + 1) * 16] = 0x38
    got, _ = run(one, s1, w, sa, M, N, K)
    v = got[5, 7] / 16.0
    print(f"W piece {piece:2d} (k-block {piece // 2 % 4}): lane scale of k-block {(np.log2(v) + 27) / 8:.2f}")
