// Epilogues and tile rasterisation shared by the MFMA contraction kernels (gemm.hip, gemm8.hip).
#pragma once
#include "gemm.h"

namespace cva {
namespace epi {

// C fragment layout: lane l, reg r -> row (l>>4)*4 + r, col l&15.  rowb/colb: this lane's first row / col.
template <typename T, int OMODE, int MI = 4, int NJ = 4>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MI][NJ], const int rowb, const int colb) {
    using TR = Traits<T>;
    T* outT = reinterpret_cast<T*>(p.out);
    float* outF = reinterpret_cast<float*>(p.out);
    if (p.dbg & 8) {   // experiment: keep the accumulators live, store one value per lane
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (rowb < p.M && colb < p.N) { if (p.out_f32) outF[(long)rowb * p.ldc + colb] = t; else outT[(long)rowb * p.ldc + colb] = TR::from_float(t); }
        return;
    }

#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = colb + j * 16;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
        // column-dependent scatter terms
        long col_term = 0; int which = 0;
        if (OMODE == OUT_QKV) {
            which = (n + p.n_off) / p.D;
            const int c = n + p.n_off - which * p.D;
            const int h = c / p.hd, d = c - h * p.hd;
            // q,k: ((s*heads+h)*L + pos)*hd + d ; vt: ((s*heads+h)*hd + d)*Lp + pos
            col_term = (which < 2 || p.v_rm) ? ((long)h * p.L * p.hd + d) : (((long)h * p.hd + d) * p.Lp);
        } else if (OMODE == OUT_CONVT) {
            const int cout = p.N >> 2;
            const int dd = n / cout, co = n - dd * cout;
            const int dy = dd >> 1, dx = dd & 1;
            col_term = ((long)dy * (2 * p.Wd) + dx) * cout + co;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = rowb + i * 16 + r;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.act == ACT_GELU) v = gelu_erf(v);
                else if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                if (OMODE == OUT_LINEAR) {
                    long orow = m;
                    if (p.o_rpi > 0) orow = (long)m + (long)(m / p.o_rpi) * p.o_extra + p.o_off;
                    if (p.res) {
                        const long rrow = p.res_mod > 0 ? (long)(m % p.res_mod) : orow;
                        v += p.res[rrow * p.ldres + n];
                    }
                    const long o = orow * (long)p.ldc + n;
                    if (p.out_f32) outF[o] = v; else outT[o] = TR::from_float(v);
                } else if (OMODE == OUT_CONVT) {
                    const int hw = p.H * p.Wd;
                    const int b = m / hw, rr = m - b * hw;
                    const int y = rr / p.Wd, x = rr - y * p.Wd;
                    const int cout = p.N >> 2;
                    const long o = (((long)b * 2 * p.H + 2 * y) * (2 * p.Wd) + 2 * x) * cout + col_term;
                    if (p.out_f32) outF[o] = v; else outT[o] = TR::from_float(v);
                } else {  // OUT_QKV
                    const int b = m / p.ntok, t = m - b * p.ntok;
                    int s = b, pos = t;
                    if (p.win > 0) {
                        const int gy = t / p.gw, gx = t - gy * p.gw;
                        const int wy = gy / p.win, wx = gx / p.win;
                        s = (b * p.nwy + wy) * p.nwx + wx;
                        pos = (gy - wy * p.win) * p.win + (gx - wx * p.win);
                    }
                    const T tv = TR::from_float(v);
                    if (which == 0) reinterpret_cast<T*>(p.q_out)[((long)s * p.heads * p.L + pos) * p.hd + col_term] = tv;
                    else if (which == 1) reinterpret_cast<T*>(p.k_out)[((long)s * p.heads * p.L + pos) * p.hd + col_term] = tv;
                    else if (p.v_rm) reinterpret_cast<T*>(p.vt_out)[((long)s * p.heads * p.L + pos) * p.hd + col_term] = tv;
                    else reinterpret_cast<T*>(p.vt_out)[(long)s * p.heads * p.hd * p.Lp + col_term + pos] = tv;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged epilogue: the 16x16 C fragments hold one column per lane, so direct stores are 2/4-byte
// scatters (one store instruction per element — store-ISSUE bound, 25-40 % of a whole GEMM).  Each wave
// transposes one 16-row slab at a time through a private LDS patch and writes 16-byte row segments.
//   st: per-wave staging area of 16 x (NJ*16 + 4) floats.  Preconditions (checked on the host, p.epi_vec):
//   N % 8 == 0 and 16-byte aligned destinations; see launch_gemm.
// ------------------------------------------------------------------------------------------------
template <typename T, int OMODE, int MI, int NJ>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& p, f32x4 (&acc)[MI][NJ], const int row0, const int col0,
                                                  float* __restrict__ st, const int lane) {
    using TR = Traits<T>;
    constexpr int WN = NJ * 16, SP = WN + 4;
    const int g = lane >> 4, li = lane & 15;
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int n = col0 + j * 16 + li; bv[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f; }
    T* outT = reinterpret_cast<T*>(p.out);
    float* outF = reinterpret_cast<float*>(p.out);

    // OUT_QKV: a 64-column wave slab lies inside one of q / k / v (D % 64 == 0)
    int which = 0;
    if (OMODE == OUT_QKV) which = (col0 + p.n_off) / p.D;
    const bool vt_slab = OMODE == OUT_QKV && which == 2 && !p.v_rm;

#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] + bv[j];
                if (p.act == ACT_GELU) v = gelu_erf(v);
                else if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                st[(g * 4 + r) * SP + j * 16 + li] = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int mrow = row0 + i * 16;
        if (vt_slab) {
            // V^T [S*heads, hd, Lp]: contiguous along the token axis -> read the slab column-wise
            const int n = col0 + lane;            // WN == 64: one column per lane
            if (n < p.N) {
                const int c = n + p.n_off - 2 * p.D;
                const int h = c / p.hd, d = c - h * p.hd;
                T* vt = reinterpret_cast<T*>(p.vt_out);
                const bool fast = p.win == 0 && (mrow + 15) < p.M && (mrow / p.ntok) == ((mrow + 15) / p.ntok) &&
                                  ((mrow % p.ntok) & 7) == 0;
                if (fast) {
                    const int b = mrow / p.ntok, t0 = mrow - b * p.ntok;
                    T* dst = vt + (((long)b * p.heads + h) * p.hd + d) * p.Lp + t0;
                    Piece pk[2 * sizeof(T) / 2];   // 16 elements: 2 pieces (fp16) or 4 pieces (fp32)
                    T* e = reinterpret_cast<T*>(pk);
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[r] = TR::from_float(st[r * SP + lane]);
#pragma unroll
                    for (int q = 0; q < (int)(2 * sizeof(T) / 2); ++q) store_piece(dst + q * (16 / (int)sizeof(T)), pk[q]);
                } else {
#pragma unroll 1
                    for (int r = 0; r < 16; ++r) {
                        const int m = mrow + r;
                        if (m >= p.M) break;
                        const int b = m / p.ntok, t = m - b * p.ntok;
                        int s_ = b, pos = t;
                        if (p.win > 0) {
                            const int gy = t / p.gw, gx = t - gy * p.gw;
                            const int wy = gy / p.win, wx = gx / p.win;
                            s_ = (b * p.nwy + wy) * p.nwx + wx;
                            pos = (gy - wy * p.win) * p.win + (gx - wx * p.win);
                        }
                        vt[(((long)s_ * p.heads + h) * p.hd + d) * p.Lp + pos] = TR::from_float(st[r * SP + lane]);
                    }
                }
            }
        } else {
            const bool f32out = (OMODE == OUT_LINEAR) && p.out_f32;
            // 8 elements per lane (T out: one 16-B store) or 4 (fp32 out: one 16-B store)
            const int epl = f32out ? 4 : 8;
            const int lpr = WN / epl;                 // lanes per row
            const int rpp = 64 / lpr;                 // rows per pass
            for (int rr = lane / lpr; rr < 16; rr += rpp) {
                const int cc = (lane - (lane / lpr) * lpr) * epl;
                const int m = mrow + rr, n = col0 + cc;
                if (m >= p.M || n >= p.N) continue;
                float v[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(st + rr * SP + cc);
                v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
                if (!f32out) {
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(st + rr * SP + cc + 4);
                    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                }
                long o;
                if (OMODE == OUT_LINEAR) {
                    long orow = m;
                    if (p.o_rpi > 0) orow = (long)m + (long)(m / p.o_rpi) * p.o_extra + p.o_off;
                    if (p.res) {
                        const long rrow = p.res_mod > 0 ? (long)(m % p.res_mod) : orow;
                        const float* rp = p.res + rrow * p.ldres + n;
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp);
                        v[0] += r0[0]; v[1] += r0[1]; v[2] += r0[2]; v[3] += r0[3];
                        if (!f32out) {
                            const f32x4 r1 = *reinterpret_cast<const f32x4*>(rp + 4);
                            v[4] += r1[0]; v[5] += r1[1]; v[6] += r1[2]; v[7] += r1[3];
                        }
                    }
                    o = orow * (long)p.ldc + n;
                    if (f32out) { f32x4 w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(outF + o) = w; continue; }
                } else if (OMODE == OUT_CONVT) {
                    const int cout = p.N >> 2;
                    const int dd = n / cout, co = n - dd * cout;
                    const int hw = p.H * p.Wd;
                    const int b = m / hw, r2 = m - b * hw;
                    const int y = r2 / p.Wd, x = r2 - y * p.Wd;
                    o = (((long)b * 2 * p.H + 2 * y + (dd >> 1)) * (2 * p.Wd) + 2 * x + (dd & 1)) * cout + co;
                } else {   // OUT_QKV, q or k slab
                    const int c = n + p.n_off - which * p.D;
                    const int h = c / p.hd, d = c - h * p.hd;
                    const int b = m / p.ntok, t = m - b * p.ntok;
                    int s_ = b, pos = t;
                    if (p.win > 0) {
                        const int gy = t / p.gw, gx = t - gy * p.gw;
                        const int wy = gy / p.win, wx = gx / p.win;
                        s_ = (b * p.nwy + wy) * p.nwx + wx;
                        pos = (gy - wy * p.win) * p.win + (gx - wx * p.win);
                    }
                    outT = reinterpret_cast<T*>(which == 0 ? p.q_out : (which == 1 ? p.k_out : p.vt_out));       // (which == 2: row-major v)
                    o = (((long)s_ * p.heads + h) * p.L + pos) * p.hd + d;
                }
                // T output: 8 elements
                if constexpr (sizeof(T) == 2) {
                    Piece pk; T* e = reinterpret_cast<T*>(&pk);
#pragma unroll
                    for (int q = 0; q < 8; ++q) e[q] = TR::from_float(v[q]);
                    if (OMODE == OUT_CONVT && !(p.dbg & 1024)) {
                        // the up-sampled activation (up to 2 GB per launch) is streamed out once and read much later:
                        // nontemporal, so it does not evict the weights / input panels
                        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                        const u32x4_t w = {pk.w[0], pk.w[1], pk.w[2], pk.w[3]};
                        __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(outT + o));
                    } else {
                        store_piece(outT + o, pk);
                    }
                } else {
                    f32x4 w0 = {v[0], v[1], v[2], v[3]}, w1 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4*>(outT + o) = w0;
                    *reinterpret_cast<f32x4*>(outT + o + 4) = w1;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


// grouped rasterisation: consecutive ids walk GM tile rows of one tile column before moving to the next
// column, so the blocks resident on one XCD at a time share few distinct A and W panels (L2 hits).
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = id / per_group;
    const int first_m = group * GM;
    const int gsz = min(GM, tiles_m - first_m);
    const int within = id - group * per_group;
    tm = first_m + within % gsz;
    tn = within / gsz;
}

}  // namespace epi
}  // namespace cva
