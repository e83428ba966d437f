// Pieces shared by the 256 x 256 x 64 fp16 contraction kernels (gemm8.hip: 8 waves, 8-phase schedule; gemm4.hip: 4 waves,
// one per SIMD, 128 x 128 accumulators per wave): LDS map, DMA helpers, and the direct epilogues on TRANSPOSED accumulator
// fragments of a 128 x 64 wave block.
#pragma once
#include "gemm.h"
#include "gemm_epilogue.h"

namespace cva {
namespace g8 {

using namespace epi;

constexpr int G8_BM = 256, G8_BN = 256, G8_BK = 64, G8_NT = 512;
constexpr int G8_TILE = 256 * 128;          // bytes of one A or W tile (256 rows x 128 B)
constexpr int G8_WOFF = 2 * G8_TILE;        // LDS layout: [E.A][O.A][E.W][O.W] -> buffer select = +32 KiB immediate offset
constexpr int G8_BIAS = 4 * G8_TILE;       // two 1-KiB bias slots (256 floats each, alternating per output tile)
constexpr int G8_LUT = 4 * G8_TILE + 2048;  // GELU: pairs (Phi(x_i), Phi'(x_i) / 128) at x_i = -8 + i/128, i = 0 .. 2048 (fp32), filled once per workgroup
constexpr int G8_LUTN = 2048;
constexpr int G8_SC = G8_LUT + (G8_LUTN + 1) * 8 + 8;       // F8: scale images, [E | O] x [A-side 1 KiB | W-side 1 KiB]
constexpr int G8_LDS = G8_SC + 4096;

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

// Pin a wave-uniform pointer into SGPRs (opaque to the optimiser): the DMA then uses the
// `global_load_lds v_off, s[base:base+1]` form instead of per-lane 64-bit induction pointers.
__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* q) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
}


// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the
// fp16 rounding of the result).  Written in erfc form so that large negative x does not cancel:
//   q = 0.5 x P(t) exp(-x^2/2), t = 1 / (1 + 0.3275911 |x| / sqrt 2);   gelu = x >= 0 ? x - q : q.
__device__ __forceinline__ float gelu_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678f, 1.0f));
    float pl = fmaf(t, 1.061405429f, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    pl *= t;
    const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504089f));
    const float q = 0.5f * x * pl * e;
    return x >= 0.f ? x - q : q;
}

// GELU on the 8-phase path: x * Phi(x) from a table in LDS, h = 1/128 over x = -8 .. 8 (|dPhi| <= h^2/8 |x phi(x)| < 1.9e-6 absolute and
// < 2e-4 relative everywhere, i.e. below half an fp16 ulp of the result).  x <= -8 -> x * 6e-16, x >= 8 -> x.  (Rounds 2-3: chord
// interpolation between the two neighbouring nodes, 7 VALU + one ds_read2_b32 per element.)
// Round 4: the same table read at the NEAREST node with the first derivative (Taylor instead of chord: the same h^2/8 |Phi''| bound,
// measured 2.3e-6 abs / 1.3e-4 rel against 2.4e-6 / 1.3e-4), two elements per instruction where a packed fp32 form exists (the
// epilogue runs no MFMAs, so the packed instructions cost nothing extra here).  u = 128 x + 1024 is rounded to the nearest integer by
// adding 1.5 * 2^23: the float's low mantissa bits ARE the node index (bits = 0x4B400000 + i), so the LDS byte address of the pair is
// one v_lshl_add_u32 of the raw bits; the fraction is 128 x + (1024 + 1.5 * 2^23 - t), exact.  Per element: 1 clamp + 1 address +
// 4 packed halves (t, c, fr, Phi) + the final multiply (half) = 4.5 VALU + one ds_read_b64, against 7 + ds_read2_b32 before.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_fill_lut(float* lut, int tid, int nthreads) {        // lut: (G8_LUTN + 1) pairs
    for (int i = tid; i <= G8_LUTN; i += nthreads) {
        const float x = -8.0f + (float)i * (1.0f / 128.f);
        lut[2 * i] = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        lut[2 * i + 1] = expf(-0.5f * x * x) * (0.3989422804014327f / 128.f);
    }
}
__device__ __forceinline__ f32x2_t gelu_lut_pair(const f32x2_t x, const unsigned lut_lds /* LDS byte address of the table */) {
    constexpr float MAGIC = 12582912.0f;                         // 1.5 * 2^23
    f32x2_t xc;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -8.0f, 7.996f); xc[1] = __builtin_amdgcn_fmed3f(x[1], -8.0f, 7.996f);
    const f32x2_t k128 = {128.f, 128.f}, kb = {1024.f + MAGIC, 1024.f + MAGIC};
    const f32x2_t t = __builtin_elementwise_fma(xc, k128, kb);   // MAGIC + round(128 x + 1024)
    const unsigned a0 = (__float_as_uint(t[0]) << 3) + (lut_lds - (0x4B400000u << 3));
    const unsigned a1 = (__float_as_uint(t[1]) << 3) + (lut_lds - (0x4B400000u << 3));
    typedef const __attribute__((address_space(3))) f32x2_t* lp;
    const f32x2_t e0 = *(lp)(unsigned long)a0, e1 = *(lp)(unsigned long)a1;       // (Phi, Phi' / 128) of the two nodes
    const f32x2_t c = kb - t;                                    // exact
    const f32x2_t fr = __builtin_elementwise_fma(xc, k128, c);   // in [-0.5, 0.5]
    const f32x2_t T = {e0[0], e1[0]}, D = {e0[1], e1[1]};
    return x * __builtin_elementwise_fma(fr, D, T);
}

// A wave-uniform divisor, opaque to the optimiser: the magic-number reciprocal of `x / d` with a loop-invariant d is otherwise hoisted out of the
// persistent tile loop, kept live in a VGPR across the K loop (which needs all 256) and spilled — every reload at a tile boundary is a scratch load
// + s_waitcnt vmcnt(0) behind the next tile's prologue DMA or the epilogue's stores (gemm8.hip g8_fresh_lane).  Re-deriving it costs ~10 VALU per tile.
__device__ __forceinline__ int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }

// Lean row loops for the hot linear flavours (fp16 out with GELU / ReLU / no activation: fc1, proj, patch, the decoder's 3x3 convolutions;
// fp32 out + fp32 residual of the same row: fc2).  The general loop of epilogue8_direct keeps ~10 run-time options alive per fragment row (row remaps with integer
// divisions, residual row modulo, output type, three activations, parked sums): measured 6.2 us per 256 x 256 tile with NO activation
// against 0.9 us of store issue (profiles/r04_m_gemm_timeline.txt) — instruction count and branch chains, not memory.  Here the options
// are decided once per tile and the row loop is straight-line code on one running pointer.
template <int ACT>
__device__ __forceinline__ void lin_rows_f16(const GemmParams& p, f32x4 (&acc)[8][4], const float (&bv)[16], const int row0, const int n,
                                             const unsigned lut_lds) {
    half_t* o = reinterpret_cast<half_t*>(p.out) + (long)row0 * p.ldc + n;
    const long step = 16L * p.ldc;
#pragma unroll
    for (int i = 0; i < 8; ++i, o += step) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2_t xin = {acc[i][j][r] + bv[j * 4 + r], acc[i][j][r + 1] + bv[j * 4 + r + 1]};
                f32x2_t y = xin;
                if (ACT == ACT_GELU) y = gelu_lut_pair(xin, lut_lds);
                if (ACT == ACT_RELU) { y[0] = fmaxf(xin[0], 0.f); y[1] = fmaxf(xin[1], 0.f); }
                v[j * 4 + r] = y[0]; v[j * 4 + r + 1] = y[1];
            }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half8_t w;
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (half_t)v[q * 8 + e];
            *reinterpret_cast<half8_t*>(o + q * 8) = w;
        }
    }
}

__device__ __forceinline__ void lin_rows_f32_res(const GemmParams& p, f32x4 (&acc)[8][4], const float (&bv)[16], const int row0, const int n) {
    float* o = reinterpret_cast<float*>(p.out) + (long)row0 * p.ldc + n;
    const float* rp = p.res + (long)row0 * p.ldres + n;
    const long step = 16L * p.ldc, rstep = 16L * p.ldres;
    f32x4 r4[2][4];                                     // the residual of row i + 1 is in flight while row i is added and stored
#pragma unroll
    for (int q = 0; q < 4; ++q) r4[0][q] = *reinterpret_cast<const f32x4*>(rp + q * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i, o += step) {
        rp += rstep;
        if (i + 1 < 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) r4[(i + 1) & 1][q] = *reinterpret_cast<const f32x4*>(rp + q * 4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 w;
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = (acc[i][q][r] + bv[q * 4 + r]) + r4[i & 1][q][r];
            *reinterpret_cast<f32x4*>(o + q * 4) = w;
        }
    }
}

// Lean rows of the fused qkv projection (q / k columns, and v when the layer keeps V row-major): token -> (sequence, position) walked 16
// tokens per fragment row with selects only.  Holds when the token count of an image and (window layers) the grid width are multiples of
// 16 — the lane's tokens m0 + li + 16 i then keep their residue mod 16, a step never skips a grid row and an image ends on a row end — and no
// token rows are padded.  The general loop below covers everything else (ViT-S: 4097 tokens).
template <bool WIN>
__device__ __forceinline__ void qkv_rows(const GemmParams& p, f32x4 (&acc)[8][4], const float (&bv)[16], half_t* qk, const long col_term,
                                         const int m_first) {
    const int ntok = opaque_s(p.ntok);
    int tb = m_first / ntok, tt = m_first - tb * ntok, tgy = 0, tgx = 0;
    const int gw = p.gw, win = opaque_s(p.win);
    const float inv_win = 1.0f / (float)(WIN ? win : 1);
    if (WIN) { const int gw_ = opaque_s(gw); tgy = tt / gw_; tgx = tt - tgy * gw_; }
    const long hl = (long)p.heads * p.L;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int s_ = tb, pos = tt;
        if (WIN) {                                      // window index by float reciprocal: exact for coordinate * win < 2^21 (caller)
            const int wy = (int)(((float)tgy + 0.5f) * inv_win), wx = (int)(((float)tgx + 0.5f) * inv_win);
            s_ = (tb * p.nwy + wy) * p.nwx + wx;
            pos = (tgy - wy * win) * win + (tgx - wx * win);
        }
        half_t* o = qk + ((long)s_ * hl + pos) * p.hd + col_term;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half8_t w;
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (half_t)(acc[i][q * 2 + (e >> 2)][e & 3] + bv[q * 8 + e]);
            *reinterpret_cast<half8_t*>(o + q * 8) = w;
        }
        tt += 16;
        if (WIN) { tgx += 16; const bool wr = tgx >= gw; tgx = wr ? tgx - gw : tgx; tgy = wr ? tgy + 1 : tgy; }
        const bool wi = tt >= ntok;                     // next image: its first grid row, column = the new token index (< 16 <= gw)
        tt = wi ? tt - ntok : tt; tb = wi ? tb + 1 : tb; tgy = wi ? 0 : tgy;
    }
}

// Direct epilogue for the TRANSPOSED accumulator orientation (C^T fragments: lane (g, li) holds, for row
// m = i*16 + li of the wave block, the 16 CONSECUTIVE columns g*16 .. g*16+15 — the W rows are permuted at DMA
// time to make them consecutive).  No LDS: bias / activation / residual on registers, 16-byte stores.
template <int OMODE, int LUT_PER_UNIT = 128>
__device__ __forceinline__ void epilogue8_direct(const GemmParams& p, f32x4 (&acc)[8][4], const float (&bv)[16],
                                                 const int mrow0, const int ncol0, const int lane, const float* lut,
                                                 const float* parked = nullptr) {
    // parked: partial sums of this lane's fragments from an earlier pass over the same tile (gemm8.hip, phase-shifted walk), f32x4
    // [fragment i * 4 + j] at a stride of 256 floats, added BEFORE bias / activation; null for ordinary tiles
    const int g = lane >> 4, li = lane & 15;
    const int n = ncol0 + g * 16;
    if (p.n_valid && n >= p.n_valid) return;            // padded columns (no cross-lane operation below: OUT_MX8 never runs padded)
    if constexpr (OMODE == OUT_LINEAR) {
        bool plain_rows = p.o_rpi <= 0 && p.res_mod <= 0 && parked == nullptr;
#ifdef CVA_ABLATION
        plain_rows = plain_rows && !(p.dbg & (512 | 1024 | 4096 | 8192 | 16384 | 65536));      // store experiments + 65536 = the general loop
#endif
        if (plain_rows) {
            const unsigned lut_lds = (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)lut;
            if (!p.res && !p.out_f32) {
                if (p.act == ACT_GELU) { lin_rows_f16<ACT_GELU>(p, acc, bv, mrow0 + li, n, lut_lds); return; }
                if (p.act == ACT_NONE) { lin_rows_f16<ACT_NONE>(p, acc, bv, mrow0 + li, n, lut_lds); return; }
                lin_rows_f16<ACT_RELU>(p, acc, bv, mrow0 + li, n, lut_lds);          // (the 3x3 convolutions of the decoder)
                return;
            } else if (p.res && p.out_f32 && p.act == ACT_NONE) {
                lin_rows_f32_res(p, acc, bv, mrow0 + li, n);
                return;
            }
        }
    }
    half_t* qk = nullptr;
    long col_term = 0;
    if (OMODE == OUT_QKV) {
        const int nn = n + p.n_off;
        const int D_ = opaque_s(p.D), hd_ = opaque_s(p.hd);
        const int which = nn / D_;
        const int c = nn - which * D_;
        const int h = c / hd_, d = c - h * hd_;             // 16 consecutive d inside one head (hd % 16 == 0)
        qk = reinterpret_cast<half_t*>(which == 0 ? p.q_out : (which == 1 ? p.k_out : p.vt_out));         // which == 2: row-major v (p.v_rm)
        col_term = (long)h * p.L * p.hd + d;
    }
    if constexpr (OMODE == OUT_QKV) {
        bool lean = !p.m_valid && (p.ntok & 15) == 0 && p.ntok >= 16;
        if (p.win > 0) lean = lean && (p.gw & 15) == 0 && p.ntok == p.gh * p.gw && p.gw <= 4096 && p.gh <= 4096 && p.win <= 256;
#ifdef CVA_ABLATION
        lean = lean && !(p.dbg & 65536);                // (A/B: the general loop)
#endif
        if (lean) {
            if (p.win > 0) qkv_rows<true>(p, acc, bv, qk, col_term, mrow0 + li);
            else qkv_rows<false>(p, acc, bv, qk, col_term, mrow0 + li);
            return;
        }
    }
    // OUT_QKV: token m -> (image b, token t, grid row gy, grid col gx), advanced by 16 tokens per fragment row without
    // integer divisions; the window index of a grid coordinate is a float reciprocal (exact: coordinate * win < 2^21)
    int tb = 0, tt = 0, tgy = 0, tgx = 0;
    const int win_o = opaque_s(p.win);                  // (opaque: 1 / win is loop invariant, see opaque_s)
    const float inv_win = 1.0f / (float)(win_o > 0 ? win_o : 1);
    const bool fwin = p.win > 0 && p.gw <= 4096 && p.gh <= 4096 && p.win <= 256;
    if (OMODE == OUT_QKV) {
        const int m_first = mrow0 + li;
        const int ntok_ = opaque_s(p.ntok), gw_ = opaque_s(p.gw);
        tb = m_first / ntok_; tt = m_first - tb * ntok_;
        if (p.win > 0) { tgy = tt / gw_; tgx = tt - tgy * gw_; }
    }
    // OUT_CONVT: input pixel m -> (image cb, row cy, column cx), likewise walked 16 pixels per fragment row: the two integer divisions per row
    // were ~560 of the ~1000 VALU instructions of this epilogue per tile and wave, next to 128-256 MFMAs for the K = 128 / 256 layers
    int cb = 0, cy = 0, cx = 0, ccout = 1, cdd = 0, cco = 0;
    if (OMODE == OUT_CONVT) {
        const int m_first = mrow0 + li, hw = opaque_s(p.H * p.Wd), wd_ = opaque_s(p.Wd);
        cb = m_first / hw;
        const int r2 = m_first - cb * hw;
        cy = r2 / wd_; cx = r2 - cy * wd_;
        ccout = opaque_s(p.N >> 2); cdd = n / ccout; cco = n - cdd * ccout;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = mrow0 + i * 16 + li;
        float v[16];
        f32x4 a[4];
        if (OMODE == OUT_LINEAR && parked) {            // (only linear launches walk phase-shifted)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = acc[i][j] + __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(parked + (i * 4 + j) * 256));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = acc[i][j];
        }
        const int act = OMODE == OUT_QKV ? (int)ACT_NONE : p.act;      // (the qkv projection has no activation)
        if (act == ACT_GELU) {                          // (uniform branches: one activation's code per launch, no selects)
            const unsigned lut_lds = (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)lut;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2_t xin = {a[j][r] + bv[j * 4 + r], a[j][r + 1] + bv[j * 4 + r + 1]};
                    const f32x2_t y = gelu_lut_pair(xin, lut_lds);
                    v[j * 4 + r] = y[0]; v[j * 4 + r + 1] = y[1];
                }
        } else if (act == ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j * 4 + r] = fmaxf(a[j][r] + bv[j * 4 + r], 0.f);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j * 4 + r] = a[j][r] + bv[j * 4 + r];
        }
        if (OMODE == OUT_LINEAR) {
            long orow = m;
            if (p.o_rpi > 0) orow = (long)m + (long)(m / opaque_s(p.o_rpi)) * p.o_extra + p.o_off;
            if (p.res) {
                const long rrow = p.res_mod > 0 ? (long)(m % opaque_s(p.res_mod)) : orow;
                const float* rp = p.res + rrow * p.ldres + n;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 r4 = *reinterpret_cast<const f32x4*>(rp + q * 4);
                    v[q * 4 + 0] += r4[0]; v[q * 4 + 1] += r4[1]; v[q * 4 + 2] += r4[2]; v[q * 4 + 3] += r4[3];
                }
            }
            if (p.out_f32) {
                float* o = reinterpret_cast<float*>(p.out) + orow * (long)p.ldc + n;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w = {v[q * 4 + 0], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]};
#ifdef CVA_ABLATION
                    if (p.dbg & 8192) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o + q * 4), "v"(w) : "memory"); continue; }
                    if (p.dbg & 16384) { __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(o + q * 4)); continue; }
#endif
                    *reinterpret_cast<f32x4*>(o + q * 4) = w;
                }
            } else {
                half_t* o = reinterpret_cast<half_t*>(p.out) + orow * (long)p.ldc + n;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    half8_t w;
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = (half_t)v[q * 8 + e];
#ifdef CVA_ABLATION      // timing experiments (wrong results): 512 = store only the first 16 bytes of every 32, 1024 = only fragment rows i < 2,
                         // 4096 = full-line pattern: instruction q writes all eight 16-byte pieces of the rows with (li & 1) == q
                    if ((p.dbg & 512) && q == 1) continue;
                    if ((p.dbg & 1024) && i >= 2) continue;
                    if (p.dbg & 4096) {
                        const int lil = lane & 15;
                        const long rsh = (long)((lil & 1) == q ? 0 : (q ? 1 : -1)) * p.ldc;     // partner row
                        *reinterpret_cast<half8_t*>(o + rsh + ((lil & 1) ? 8 : 0)) = w;
                        continue;
                    }
                    if (p.dbg & 8192) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o + q * 8), "v"(w) : "memory"); continue; }
                    if (p.dbg & 16384) { __builtin_nontemporal_store(w, reinterpret_cast<half8_t*>(o + q * 8)); continue; }
#endif
                    *reinterpret_cast<half8_t*>(o + q * 8) = w;
                }
            }
        } else if (OMODE == OUT_MX8) {
            // MX-fp8 output (the hidden activation of the MLP, consumed as the A operand of fc2): a 32-column scale block is
            // this lane's 16 values + those of lane ^ 16 (g ^ 1).  OCP MX: shared exponent = floor(log2(amax)) - 8 (e4m3 emax),
            // elements = v * 2^-shared, saturated to +-448, round to nearest even (v_cvt_pk_fp8_f32).
            float amax = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(v[e]));
            amax = fmaxf(amax, __shfl_xor(amax, 16));
            const int ex = (int)((__float_as_uint(amax) >> 23) & 0xff);          // biased exponent of amax (0 for zero / denormal)
            int sbyte = ex - 8; sbyte = sbyte < 0 ? 0 : sbyte;                  // E8M0 byte = shared exponent + 127
            const float inv = __uint_as_float((unsigned)(254 - sbyte) << 23);   // 2^-(sbyte - 127)
            i32x4_t w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int pk = 0;
                const float x0 = __builtin_amdgcn_fmed3f(v[q * 4 + 0] * inv, -448.f, 448.f), x1 = __builtin_amdgcn_fmed3f(v[q * 4 + 1] * inv, -448.f, 448.f);
                const float x2 = __builtin_amdgcn_fmed3f(v[q * 4 + 2] * inv, -448.f, 448.f), x3 = __builtin_amdgcn_fmed3f(v[q * 4 + 3] * inv, -448.f, 448.f);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, pk, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, pk, true);
                w[q] = pk;
            }
            unsigned char* o8 = reinterpret_cast<unsigned char*>(p.out) + (long)m * p.ldc + n;
            *reinterpret_cast<i32x4_t*>(o8) = w;
            if (!(g & 1))
                reinterpret_cast<unsigned char*>(p.out_scale)[mx8_scale_index(m, n, p.N, false)] = (unsigned char)sbyte;
        } else if (OMODE == OUT_CONVT) {
            // ConvTranspose2d k2 s2: row m = input pixel (b, y, x), column n = (dy*2 + dx) * Cout + co; the lane's 16 columns
            // are 16 consecutive co of one (dy, dx) (Cout % 16 == 0): one 32-byte run of output pixel (2y + dy, 2x + dx).
            // The up-sampled activation is streamed out once and read much later: nontemporal stores.
            const int cout = ccout, dd = cdd, co = cco;
            const int b = cb, y = cy, x = cx;
            cx += 16;                                   // next fragment row: 16 input pixels further (raster order over images)
            while (cx >= p.Wd) { cx -= p.Wd; if (++cy >= p.H) { cy = 0; ++cb; } }
            half_t* o = reinterpret_cast<half_t*>(p.out) +
                        (((long)b * 2 * p.H + 2 * y + (dd >> 1)) * (2 * p.Wd) + 2 * x + (dd & 1)) * cout + co;
            if (p.comp_bias) {
                // composed ConvTranspose2d o Conv2d 3x3: the 3x3 taps that fall outside the 2H x 2W image contribute neither their
                // products (the input pixel's tap is masked in the A tile) nor their share of the transposed convolution's bias —
                // border pixels take their bias from the per-case table instead of the staged interior one
                const int Y = 2 * y + (dd >> 1), X = 2 * x + (dd & 1);
                const int rc = Y == 0 ? 0 : (Y == 2 * p.H - 1 ? 2 : 1), cc = X == 0 ? 0 : (X == 2 * p.Wd - 1 ? 2 : 1);
                if (rc != 1 || cc != 1) {
                    const float* tb_ = p.comp_bias + (long)(rc * 3 + cc) * cout + co;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(tb_ + q * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = acc[i][q][r] + t4[r];
                            v[q * 4 + r] = p.act == ACT_RELU ? fmaxf(a, 0.f) : a;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {          // consumed by the next decoder stage: ordinary stores
                    half8_t w;
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = (half_t)v[q * 8 + e];
                    *reinterpret_cast<half8_t*>(o + q * 8) = w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    half8_t w;
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = (half_t)v[q * 8 + e];
                    __builtin_nontemporal_store(w, reinterpret_cast<half8_t*>(o + q * 8));
                }
            }
        } else {   // OUT_QKV, q or k columns
            int s_ = tb, pos = tt;
            if (p.win > 0) {
                const int wy = fwin ? (int)(((float)tgy + 0.5f) * inv_win) : tgy / p.win;
                const int wx = fwin ? (int)(((float)tgx + 0.5f) * inv_win) : tgx / p.win;
                s_ = (tb * p.nwy + wy) * p.nwx + wx;
                pos = (tgy - wy * p.win) * p.win + (tgx - wx * p.win);
            }
            // next fragment row: 16 tokens further
            tt += 16; tgx += 16;
            if (tt >= p.ntok) {
                tt -= p.ntok; ++tb;
                if (p.win > 0) { const int gw_ = opaque_s(p.gw); tgy = tt / gw_; tgx = tt - tgy * gw_; }
            } else if (p.win > 0) {
                while (tgx >= p.gw) { tgx -= p.gw; ++tgy; }
            }
            half_t* o = qk + ((long)s_ * p.heads * p.L + pos) * p.hd + col_term;
            if (p.m_valid && m >= p.m_valid) continue;      // padded rows (the token walk above is already advanced)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                half8_t w;
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = (half_t)v[q * 8 + e];
                *reinterpret_cast<half8_t*>(o + q * 8) = w;
            }
        }
    }
}


// Direct V^T epilogue for the v columns of the fused qkv projection.  Those tiles run the SAME transposed
// kernel with the operands exchanged (C^T = W_v . X^T: the "A" tile holds 256 W rows, the permuted "W" tile 256
// token rows), so lane (g, li) holds, for column n = i*16 + li of the wave block, the 16 CONSECUTIVE tokens
// g*16 .. g*16+15.  V^T is [S*heads, hd, Lp], contiguous along the token position: global attention gets
// 16-byte stores, window-partitioned layers one 16-byte + 8- / 4-byte stores per run (grid width % 16 == 0) or, in general,
// 4-byte stores of token pairs (a pair never straddles a window when the window size and the grid width are even).
__device__ __forceinline__ void epilogue8_vt(const GemmParams& p, f32x4 (&acc)[8][4], const float (&bvt)[16],
                                             const int nrow0, const int mcol0, const int lane) {
    const int g = lane >> 4, li = lane & 15;
    const int mb = mcol0 + g * 16;                               // first of this lane's 16 tokens
    half_t* vt = reinterpret_cast<half_t*>(p.vt_out);
    const int ntok_ = opaque_s(p.ntok);
    const int b0 = mb / ntok_, t0 = mb - b0 * ntok_;
    const int mlim = p.m_valid ? p.m_valid : 0x7fffffff;        // padded token rows are dropped
    if (mb >= mlim) return;
    const bool fast = p.win == 0 && t0 + 15 < p.ntok && (t0 & 7) == 0 && mb + 15 < mlim;
    // Window layers whose grid width is a multiple of 16 (1024-px tiles: 64): the lane's 16 tokens lie in ONE grid row and
    // split into at most two runs that are contiguous in V^T — the rest of window wxA's row (lenA tokens from px = pxA) and
    // the start of the next window's row.  With even window size all lengths are even: dwords 0-3 go out as one 16-byte
    // store, dwords 4-7 as 8-byte stores where the pair stays inside a run and 4-byte stores where it straddles the window
    // edge — 3.5 stores per lane and row on average instead of 8 (every store instruction of this epilogue touches 64
    // different cache lines, so the count is the cost: the window layers' qkv launch was 24 % slower than the global ones').
    bool wide = false;
    long offA = 0, offB = 0; int nA = 8;
    if (!fast && p.win >= 8 && (p.win & 1) == 0 && (p.gw & 15) == 0 && p.ntok == p.gh * p.gw && (t0 & 15) == 0) {
        const int gy = t0 / p.gw, gx0 = t0 - gy * p.gw;
        const int wy = gy / p.win, py = gy - wy * p.win;
        const int wxA = gx0 / p.win, pxA = gx0 - wxA * p.win;
        const int lenA = min(16, p.win - pxA);
        nA = lenA >> 1;
        const long sA = ((long)b0 * p.nwy + wy) * p.nwx + wxA;
        offA = sA * p.heads * p.hd * p.Lp + py * p.win + pxA;
        offB = (sA + 1) * p.heads * p.hd * p.Lp + py * p.win;
        wide = __all(lenA >= 8 && (pxA & 1) == 0) != 0;          // wave-uniform: every lane's run A holds the 16-byte store
    }
    // token -> offset inside one (s, h, d) row of V^T for the 16 tokens (8 pairs), walked incrementally (no divisions
    // per token; window index by float reciprocal, exact for coordinate * win < 2^21)
    long po0[8], po1[8]; bool pair_ok[8];
    unsigned tok_ok = 0xffffu;                                   // bit 2u + w: token mb + 2u + w is a real row
    if (!fast && !wide) {
        const float inv_win = 1.0f / (float)(p.win > 0 ? p.win : 1);
        const bool fwin = p.win > 0 && p.gw <= 4096 && p.gh <= 4096 && p.win <= 256;
        int b = b0, t = t0, gy = 0, gx = 0;
        if (p.win > 0) { gy = t / p.gw; gx = t - gy * p.gw; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            long o[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                int s_ = b, pos = t;
                if (p.win > 0) {
                    const int wy = fwin ? (int)(((float)gy + 0.5f) * inv_win) : gy / p.win;
                    const int wx = fwin ? (int)(((float)gx + 0.5f) * inv_win) : gx / p.win;
                    s_ = (b * p.nwy + wy) * p.nwx + wx;
                    pos = (gy - wy * p.win) * p.win + (gx - wx * p.win);
                }
                o[w] = (long)s_ * p.heads * p.hd * p.Lp + pos;
                ++t; ++gx;
                if (t >= p.ntok) { t = 0; ++b; gy = 0; gx = 0; }
                else if (p.win > 0 && gx >= p.gw) { gx = 0; ++gy; }
            }
            po0[u] = o[0]; po1[u] = o[1];
            pair_ok[u] = (o[1] == o[0] + 1) && ((o[0] & 1) == 0) && mb + 2 * u + 1 < mlim;
            if (mb + 2 * u >= mlim) tok_ok &= ~(1u << (2 * u));
            if (mb + 2 * u + 1 >= mlim) tok_ok &= ~(2u << (2 * u));
        }
    }
    const int c0 = nrow0 + li + p.n_off - 2 * p.D;              // v column of fragment row 0; +16 per fragment row
    const int hd_ = opaque_s(p.hd);
    int h = c0 / hd_, d = c0 - h * hd_;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float bv = bvt[i];
        const long rowoff = ((long)h * p.hd + d) * p.Lp;
        d += 16;
        while (d >= p.hd) { d -= p.hd; ++h; }
        if (p.n_valid && c0 + 16 * i >= p.D) continue;           // padded v columns
        if (fast) {
            half_t* dst = vt + (long)b0 * p.heads * p.hd * p.Lp + rowoff + t0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                half8_t w;
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = (half_t)(acc[i][q * 2 + (e >> 2)][e & 3] + bv);
                *reinterpret_cast<half8_t*>(dst + q * 8) = w;
            }
        } else if (wide) {
            typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
            typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
            half2_t dw[8];                                        // dword k = tokens 2k, 2k+1
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                dw[k][0] = (half_t)(acc[i][k >> 1][(k & 1) * 2] + bv);
                dw[k][1] = (half_t)(acc[i][k >> 1][(k & 1) * 2 + 1] + bv);
            }
            half_t* rA = vt + offA + rowoff;
            half_t* rB = vt + offB + rowoff - 2 * nA;             // run B holds dwords nA .. 7: dword k at rB + 2k
            {
                half8_t w;
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = dw[e >> 1][e & 1];
                *reinterpret_cast<half8_t*>(rA) = w;
            }
            // dwords (4,5): both in A (nA >= 6), both in B (nA == 4), or split (nA == 5)
            if (nA != 5) {
                const half4_t w = {dw[4][0], dw[4][1], dw[5][0], dw[5][1]};
                *reinterpret_cast<half4_t*>((nA >= 6 ? rA : rB) + 8) = w;
            } else {
                *reinterpret_cast<half2_t*>(rA + 8) = dw[4];
                *reinterpret_cast<half2_t*>(rB + 10) = dw[5];
            }
            // dwords (6,7): both in A (nA == 8), both in B (nA <= 6), or split (nA == 7)
            if (nA != 7) {
                const half4_t w = {dw[6][0], dw[6][1], dw[7][0], dw[7][1]};
                *reinterpret_cast<half4_t*>((nA == 8 ? rA : rB) + 12) = w;
            } else {
                *reinterpret_cast<half2_t*>(rA + 12) = dw[6];
                *reinterpret_cast<half2_t*>(rB + 14) = dw[7];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const half_t v0 = (half_t)(acc[i][u >> 1][(u & 1) * 2] + bv), v1 = (half_t)(acc[i][u >> 1][(u & 1) * 2 + 1] + bv);
                if (pair_ok[u]) {
                    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
                    const half2_t w = {v0, v1};
                    *reinterpret_cast<half2_t*>(vt + po0[u] + rowoff) = w;
                } else {
                    if (tok_ok & (1u << (2 * u))) vt[po0[u] + rowoff] = v0;
                    if (tok_ok & (2u << (2 * u))) vt[po1[u] + rowoff] = v1;
                }
            }
        }
    }
}

}  // namespace g8
}  // namespace cva
