// Single-pass attention for SHORT key sequences (<= 208 keys: the 14x14 = 196-position windows of the SAM
// encoder, SAM/image_encoder.py:128-193, 235-257), fp16 storage, gfx950.
//
// With at most 13 key blocks of 16 the whole score row of a query fits in registers (2 query blocks x 13 key
// blocks x 4 = 104 accumulator VGPRs per lane), so the softmax is ONE pass: no running max / sum, no rescale of
// the output accumulators, no per-tile cross-lane reductions, and the ragged tail (196 = 12 x 16 + 4) costs one
// masked key block instead of a whole 64-key tile.  Orientation as in attention2.hip (transposed flash attention):
//   S^T = K . Q^T  -> lane (g, li) holds keys kb*16 + g*4 + r of ONE query (li): row statistics are lane-local
//                     plus two shuffles;
//   O^T = V^T . P^T with P^T taken straight from the S^T registers (relabelled contraction index).
// One workgroup = 4 waves of one (window, head), which walk its queries in (up to) two passes of 4 x 32.  K passes through LDS in 64-key tiles (all fetched up front),
// V^T (all keys) and the one-hot rel-pos matrix E[key][kh | KH + kw] (all keys) are staged once per workgroup.
// The decomposed rel-pos bias (image_encoder.py:354-392) is one extra contraction step of the S^T MFMA against E,
// with the per-query terms G = Q . tab^T computed by MFMA at block start (tables staged through LDS).
#include "attention.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace cva {

namespace {

constexpr int WKT = 64, WNT = 256, WQW = 32, WQT = 128, WNKB = 13, WKEYS = WNKB * 16;   // 208 keys max
constexpr float W_LOG2E = 1.4426950408889634f;

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ half8_t frag2x4(const half_t* p0, const half_t* p1) {
    const half4_t a = *reinterpret_cast<const half4_t*>(p0);
    const half4_t b = *reinterpret_cast<const half4_t*>(p1);
    half8_t f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
    return f;
}

template <int HD, int BIAS>
__global__ __launch_bounds__(WNT, 2) void attnw_kernel(const AttnParams p) {
    using TR = Traits<half_t>;
    constexpr int PE = 8;
    constexpr int HDP = (HD + 31) / 32 * 32, NKS = HDP / 32, ND = HD / 16;
    constexpr int PK = lds_pitch<half_t>(HDP);          // K tile / table row pitch
    constexpr int PVF = WKEYS + 4;                      // V^T row pitch 424 B = 106 dwords: the 16 rows of a lane group fall on 16 distinct
                                                        // bank pairs under the 32-bank map of ds_read2_b64 (the compiler merges the two 8-byte
                                                        // halves of a fragment read) AND the 64-bank map; rows are 8-byte aligned only
    constexpr int PE1 = 32 + 8;                         // E / relcat row pitch: 80 B (an odd multiple of 16 B — two-way conflicts for b128 fragment reads
                                                        // under gfx950's lane grouping, see AttnwpGeom; this kernel serves the non-production shapes)

    extern __shared__ __attribute__((aligned(16))) unsigned char smemw[];
    half_t* Ks = reinterpret_cast<half_t*>(smemw);      // [64][PK]   (also the staging area of the rel-pos tables)
    half_t* Vts = Ks + WKT * PK;                        // [HD][PVF]
    half_t* Es = Vts + HD * PVF;                        // BIAS: [208][PE1]
    half_t* Rc = Es + (BIAS ? WKEYS * PE1 : 0);         // BIAS: [128][PE1]

    if (p.dbg & 8) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
        // one workgroup per (sequence, head): its (up to) two passes of 128 queries share the staged V^T / E / K tiles
    const int sh = xcd_remap(blockIdx.x, gridDim.x);
    const int npass = p.L > WQT ? 2 : 1;                // block-uniform

    const half_t* __restrict__ Qg = reinterpret_cast<const half_t*>(p.Q) + (long)sh * p.L * HD;
    const half_t* __restrict__ Kg = reinterpret_cast<const half_t*>(p.K) + (long)sh * p.L * HD;
    const half_t* __restrict__ Vg = reinterpret_cast<const half_t*>(p.Vt) + (long)sh * HD * p.Lp;

    // ---- Q fragments (B operand of S^T): pass ps covers queries ps*128 + wave*32 + qb*16 + li
    half8_t qf[2][NKS];
    auto load_q = [&](int ps) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int row = ps * WQT + wave * WQW + qb * 16 + li;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int d0 = ks * 32 + g * 8;
                qf[qb][ks] = (row < p.L && d0 < HD) ? *reinterpret_cast<const half8_t*>(Qg + (long)row * HD + d0) : (half8_t)(0);
            }
        }
    };
    load_q(0);

    // ---- K: all (<= 4) tiles of 64 keys are fetched into registers up front (48 VGPRs for hd 80): the key loop below then
    // contains no global-memory round trip at all
    constexpr int KPPR = HD / PE;
    constexpr int KN = (WKT * KPPR + WNT - 1) / WNT;
    Piece kreg[4][KN];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < KN; ++u) {
            const int i = tid + u * WNT;
            const int r = i / KPPR, c = i - r * KPPR;
            const int key = t * WKT + r;
            kreg[t][u] = (i < WKT * KPPR && key < p.nk && !(p.dbg & 1)) ? load_piece(Kg + (long)key * HD + c * PE) : zero_piece();
        }

    // ---- V^T, all keys, once (elements past the last key are zeroed: they meet P = 0 but must not be NaN).
    // All global loads of the prologue (Q, K tile 0, V^T, rel-pos tables) are issued back to back into registers
    // before the first LDS write, so the workgroup pays ONE memory round trip instead of one per loop iteration.
    constexpr int VPPR = WKEYS / PE;                    // 26 pieces per row
    constexpr int VN = (HD * VPPR + WNT - 1) / WNT;
    Piece vreg[VN];
#pragma unroll
    for (int u = 0; u < VN; ++u) {
        const int i = tid + u * WNT;
        const int d = i / VPPR, c = i - d * VPPR;
        const int key0 = c * PE;
        vreg[u] = (i < HD * VPPR && key0 < p.nk && !(p.dbg & 1)) ? load_piece(Vg + (long)d * p.Lp + key0) : zero_piece();
    }
    // rel-pos operands prepared once per layer by attnw_prep_kernel (identical for every workgroup, L2 resident):
    //   prep[0 .. 208*32)            one-hot rows E[key][kh | KH + kw]           (fp16)
    //   prep[208*32 .. + 2*32*HDP)   the two tables as zero-padded fp16 rows     (fp16 [2][32][HDP])
    const half_t* __restrict__ prepE = reinterpret_cast<const half_t*>(p.win_prep);
    const half_t* __restrict__ prepT = prepE + WKEYS * 32;
    constexpr int EN = (WKEYS * 4 + WNT - 1) / WNT;
    Piece ereg[BIAS ? EN : 1];
    half8_t tf[BIAS ? 4 : 1][NKS];
    if (BIAS) {
#pragma unroll
        for (int u = 0; u < EN; ++u) {
            const int i = tid + u * WNT;
            ereg[u] = i < WKEYS * 4 ? load_piece(prepE + i * PE) : zero_piece();
        }
    }
    auto load_tf = [&]() {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)                  // tj = tbl*2 + jb: table rows tbl*32 + jb*16 + li
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                tf[tj][ks] = *reinterpret_cast<const half8_t*>(prepT + ((tj >> 1) * 32 + (tj & 1) * 16 + li) * HDP + ks * 32 + g * 8);
    };
    if (BIAS) load_tf();
    // ---- LDS writes: V^T (all keys), E, K tile 0 (+ zero pad columns of the K rows)
#pragma unroll
    for (int u = 0; u < VN; ++u) {
        const int i = tid + u * WNT;
        if (i < HD * VPPR) {
            const int d = i / VPPR, c = i - d * VPPR;
            const int key0 = c * PE;
            Piece v = vreg[u];
            if (key0 + PE > p.nk) {
                half_t* e = reinterpret_cast<half_t*>(&v);
#pragma unroll
                for (int j = 0; j < PE; ++j) if (key0 + j >= p.nk) e[j] = (half_t)0.f;
            }
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(Vts + d * PVF + c * PE);   // two 8-byte stores
            dst[0] = ((unsigned long long)v.w[1] << 32) | v.w[0];
            dst[1] = ((unsigned long long)v.w[3] << 32) | v.w[2];
        }
    }
    if (HDP > HD) {
        for (int i = tid; i < WKT * (HDP - HD) / PE; i += WNT) {
            const int r = i / ((HDP - HD) / PE), c = i - r * ((HDP - HD) / PE);
            store_piece(Ks + r * PK + HD + c * PE, zero_piece());
        }
    }
#pragma unroll
    for (int u = 0; u < KN; ++u) {
        const int i = tid + u * WNT;
        if (i < WKT * KPPR) { const int r = i / KPPR, c = i - r * KPPR; store_piece(Ks + r * PK + c * PE, kreg[0][u]); }
    }
    if (BIAS) {
#pragma unroll
        for (int u = 0; u < EN; ++u) {
            const int i = tid + u * WNT;
            if (i < WKEYS * 4) store_piece(Es + (i >> 2) * PE1 + (i & 3) * PE, ereg[u]);
        }
    }
    __syncthreads();                                    // V^T / E / K tile 0 visible

    const int ntiles = (p.nk + WKT - 1) / WKT;          // <= 4
    const int nkb_all = (p.nk + 15) / 16;               // <= 13
    const float c1 = p.scale * W_LOG2E;
    const int s_idx = sh / p.heads, h = sh - s_idx * p.heads;
    half_t* __restrict__ out = reinterpret_cast<half_t*>(p.out);

    auto run_pass = [&](auto passc) {
        constexpr int PS = decltype(passc)::value;
        const int q0 = PS * WQT + wave * WQW;
        const bool wave_active = q0 < p.L && !(p.dbg & 2);
        if (PS > 0 && wave_active && BIAS) load_tf();       // (Q of this pass was requested at the end of the previous pass' S loop)

        // ---- relcat[q][kh] = q . tab_h[qy - kh + KH - 1] / scale ; relcat[q][KH + kw] likewise (image_encoder.py:347-351);
        // each wave fills and reads back only ITS rows of Rc (LDS is in order per wave: no barrier)
        half8_t bf[2];
        if (BIAS && wave_active) {
            half_t* myrc = Rc + (wave * WQW) * PE1;
            for (int i = lane; i < WQW * PE1 / PE; i += 64) store_piece(myrc + i * PE, zero_piece());
            const float inv_scale = 1.0f / p.scale;
#pragma unroll
            for (int tbl = 0; tbl < 2; ++tbl) {
                const int Ksz = tbl == 0 ? p.KH : p.KW;
                const int off = tbl == 0 ? 0 : p.KH;
                const int nj = 2 * Ksz - 1;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        f32x4 acc = (f32x4)(0.f);
#pragma unroll
                        for (int ks = 0; ks < NKS; ++ks)
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(tf[tbl * 2 + jb][ks], qf[qb][ks], acc, 0, 0, 0);
                        const int q = q0 + qb * 16 + li;
                        if (q < p.L) {
                            const int qy = q / p.KW, qx = q - qy * p.KW;
                            const int c = tbl == 0 ? qy : qx;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int jj = jb * 16 + g * 4 + r;
                                const int kk = c - jj + Ksz - 1;
                                if (jj < nj && kk >= 0 && kk < Ksz) myrc[(qb * 16 + li) * PE1 + off + kk] = (half_t)(acc[r] * inv_scale);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) bf[qb] = *reinterpret_cast<const half8_t*>(myrc + (qb * 16 + li) * PE1 + g * 8);
        }

        // ---- S^T for all keys: s[qb][kg][r] = score(query li of qb, key kg*16 + g*4 + r) / scale
        f32x4 s[2][WNKB];
#pragma unroll
        for (int kg = 0; kg < WNKB; ++kg) { s[0][kg] = (f32x4)(0.f); s[1][kg] = (f32x4)(0.f); }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < ntiles) {                               // block-uniform
                if (t > 0 || PS > 0) {                      // (tile 0 of pass 0 was staged by the prologue)
                    __syncthreads();                        // every wave is done reading the previous tile
#pragma unroll
                    for (int u = 0; u < KN; ++u) {
                        const int i = tid + u * WNT;
                        if (i < WKT * KPPR) { const int r = i / KPPR, c = i - r * KPPR; store_piece(Ks + r * PK + c * PE, kreg[t][u]); }
                    }
                    __syncthreads();
                }
                if (wave_active) {
                    // k-step outermost: the MFMAs of one k-step write different accumulators (dependent MFMAs 8 apart)
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
                            const int kg = t * 4 + kb;
                            if (kg < WNKB && kg < nkb_all) {
                                const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (kb * 16 + li) * PK + ks * 32 + g * 8);
                                s[0][kg < WNKB ? kg : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[0][ks], s[0][kg < WNKB ? kg : 0], 0, 0, 0);
                                s[1][kg < WNKB ? kg : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[1][ks], s[1][kg < WNKB ? kg : 0], 0, 0, 0);
                            }
                        }
                    }
                    if (BIAS) {
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
                            const int kg = t * 4 + kb;
                            if (kg < WNKB && kg < nkb_all) {
                                const half8_t ef = *reinterpret_cast<const half8_t*>(Es + (kg * 16 + li) * PE1 + g * 8);
                                s[0][kg < WNKB ? kg : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ef, bf[0], s[0][kg < WNKB ? kg : 0], 0, 0, 0);
                                s[1][kg < WNKB ? kg : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ef, bf[1], s[1][kg < WNKB ? kg : 0], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        if (!wave_active) return;                           // (the barriers of a following pass are outside this lambda's tail)
        if (PS == 0 && npass > 1) load_q(1);                // Q of the next pass: in flight during this pass' softmax and PV

        // ---- one-pass softmax in the log2 domain, P^T fragments, O^T = V^T . P^T
        f32x4 o[2][ND];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int n = 0; n < ND; ++n) o[qb][n] = (f32x4)(0.f);
        float inv_l[2];
        half8_t pf[2][7];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kg = 0; kg < WNKB; ++kg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kg * 16 + g * 4 + r;
                    const float v = key < p.nk ? s[qb][kg][r] * c1 : -INFINITY;
                    s[qb][kg][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kg = 0; kg < 14; ++kg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = 0.f;
                    if (kg < WNKB) { pv = __builtin_amdgcn_exp2f(s[qb][kg < WNKB ? kg : 0][r] - mx); sum += pv; }
                    // contraction slot (m, g*8 + j): j < 4 -> key block 2m, reg j ; j >= 4 -> key block 2m+1, reg j-4
                    pf[qb][kg >> 1][(kg & 1) * 4 + r] = (half_t)pv;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            inv_l[qb] = 1.0f / sum;
        }
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            if (2 * m < nkb_all) {                          // wave-uniform: all P of these 32 keys are exactly 0 otherwise
#pragma unroll
                for (int n = 0; n < ND; ++n) {
                    const half_t* vrow = Vts + (n * 16 + li) * PVF + g * 4;
                    // key blocks 2m and 2m+1; block 13 does not exist (its P is 0): read block 12 again instead of past the row
                    const half8_t vf = frag2x4(vrow + (2 * m) * 16, vrow + (m == 6 ? 2 * m : 2 * m + 1) * 16);
                    o[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[0][m], o[0][n], 0, 0, 0);
                    o[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[1][m], o[1][n], 0, 0, 0);
                }
            }
        }

        // ---- normalise; lane holds O[query li of qb][d = n*16 + g*4 + r]
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qg = q0 + qb * 16 + li;
            if (qg >= p.L || ((p.dbg & 4) && o[qb][0][0] != 12345.f)) continue;
            long row;
            if (p.win > 0) {
                const int nw = p.nwx * p.nwy;
                const int b = s_idx / nw, w = s_idx - b * nw;
                const int wy = w / p.nwx, wx = w - wy * p.nwx;
                const int py = qg / p.win, px = qg - py * p.win;
                const int gy = wy * p.win + py, gx = wx * p.win + px;
                if (gy >= p.gh || gx >= p.gw) continue;
                row = (long)b * p.ntok + gy * p.gw + gx;
            } else {
                row = (long)s_idx * p.ntok + qg;
            }
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                half4_t v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[qb][n][r] * inv_l[qb]);
                *reinterpret_cast<half4_t*>(out + row * p.D + h * HD + n * 16 + g * 4) = v;
            }
        }
    };
    run_pass(std::integral_constant<int, 0>{});
    if (npass > 1) run_pass(std::integral_constant<int, 1>{});
}


// ---- persistent variant for the two-pass case (128 < L <= 208: the 196-position SAM windows) -----------------------
// One workgroup of 8 waves per CU walks (window, head) items grid-stride; wave w owns queries 32w .. 32w+31, so an
// item is ONE pass.  K of item i+1 is DMA'd (global_load_lds) into the second K image and its V^T pieces are requested
// into registers right after item i's LDS image is complete, so both land while item i computes; Q of item i+1 is
// requested when item i's S loop has consumed its Q fragments.  Every wave retires its own loads (vmcnt(0)) BEFORE it
// issues the item's output stores, so store drain is never waited for.  E and the fp16 tables are staged once per
// workgroup.  All of K is resident in LDS (pitch hd + 8: conflict-free b128 fragment reads), so the key loop has no
// barriers: two barriers per item in total.
constexpr int PNT = 512;

__device__ __forceinline__ const unsigned char* uniform_ptr_w(const void* q) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
}
// global -> LDS DMA of 64 x 16 B; M0 = LDS byte address of the 1-KiB destination (no other code here depends on M0)
#define AW_DMA(voff, base, ldsaddr)                                                                         \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), \
                 "s"(ldsaddr) : "memory")

#define AW_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define AW_DSR2(dst, addr, o0, o1) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"(o1))
#define AW_WAIT(n)                                                                                          \
    do {                                                                                                    \
        half8_t &r0_ = ring[0], &r1_ = ring[1], &r2_ = ring[2], &r3_ = ring[3];                            \
        asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r0_), "+v"(r1_), "+v"(r2_), "+v"(r3_) :: "memory"); \
    } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int HD>
struct AttnwpGeom {
    static constexpr int HDP = (HD + 31) / 32 * 32;
    static constexpr int KPPR = HD / 8;                 // pieces per K row in memory
    // Row pitches of everything the S^T phase reads with ds_read_b128 (K image, E, the tables): a multiple of 64 B plus 32.  gfx950 serves a b128 read in four
    // groups of sixteen lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) — and with a lane's row = lane & 15 and its 16-byte
    // column = lane >> 4 only those pitches put a group's sixteen 16-byte accesses on sixty-four distinct banks: 160 B (hd 80: the rows as they lie in memory),
    // 96 B, 224 B take 4 LDS cycles per wave-instruction; the "odd multiple of 16 B" pitches of rounds 2 - 4 (176 B, 80 B, 208 B) take 8.
    static constexpr int PKP = (HD * 2) % 64 == 32 ? HD : HD + 16;      // K row pitch in halves: 80 (hd 80), 80 (hd 64)
    static constexpr int LPPR = PKP / 8;                // pieces per K row in LDS; pieces past KPPR repeat the last one (finite data against zero Q fragments)
    static constexpr int NDMA = (WKEYS * LPPR + 63) / 64;
    static constexpr int KROWS = (NDMA * 64 + LPPR - 1) / LPPR;          // > WKEYS: the last k-step of hd 80 reads 32 bytes into the NEXT row
    static constexpr int KIMG = KROWS * PKP;            // halves per K image
    static constexpr int PE1 = 48;                      // E row pitch (96 B)
    static constexpr int PT = (HDP * 2) % 64 == 32 ? HDP : HDP + 16;    // table row pitch: 112 halves = 224 B (hd 80), 80 halves = 160 B (hd 64)
    static constexpr int PWR = 48;                      // halves per row of the kw-term staging area (see the kernel: skewed, unconditional writes)
    static constexpr size_t lds_bytes() {
        return (size_t)(2 * KIMG + HD * (WKEYS + 4) + WKEYS * PE1 + (PNT / 64) * 32 * PWR + 64 * PT) * sizeof(half_t);
    }
};

// VRM != 0 (AttnParams::v_rm, hd 80): V arrives ROW-major [L][HD] like K.  Its LDS image is then the item's V rows as they lie in
// memory (208 rows x 160 B, linear), and the PV operand comes out of ds_read_b64_tr_b16 (see attention2.hip).
//   VRM = 2 (production): V by LDS-DMA like K — no register staging, no LDS stores by the waves.  Only ONE V image fits next to the two K
//     images, so item i's V is requested at the top of item i (the image was released by the barrier that ends item i-1), has the
//     relcat + S^T phase to land and is published by a barrier behind S^T (two barriers per item, as before);
//   VRM = 1 (ablation builds, CVA_VRM_REG=1): the pieces of item i+1's V are requested into registers while item i computes and written
//     to the image with five ds_write_b128 per thread at the top of item i+1 (the structure of the V^T form).
//   Same box, same call (profiles/r04_e_vrm_window_variants.txt): V^T 1085.5 us per launch of 64 tiles, VRM 1 1052.2, VRM 2 1022.6.
typedef __fp16 w_fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ half8_t w_frag_tr_2x4(const half_t* p0, const half_t* p1) {
    typedef __attribute__((address_space(3))) w_fp16x4_t* lp;
    const half4_t a = __builtin_bit_cast(half4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(__attribute__((address_space(3))) void*)const_cast<half_t*>(p0)));
    const half4_t b = __builtin_bit_cast(half4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(__attribute__((address_space(3))) void*)const_cast<half_t*>(p1)));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// DBG (ablation library only, CVA_ATTNWP_DBG, VRM = 2; results wrong by construction): 1 no rel-pos phase, 2 no S^T MFMAs / fragment reads, 4 no PV MFMAs / fragment
// reads, 8 no softmax arithmetic, 16 no K / V DMA, 32 no output stores — what the item's time is made of (profiles/r05_o_attnwp_ablation.txt).
template <int HD, int BIAS, int VRM = 0, int DBG = 0>
__global__ __launch_bounds__(PNT) void attnwp_kernel(const AttnParams p) {
    static_assert(!VRM || HD == 80, "row-major V: hd 80");
    using GM = AttnwpGeom<HD>;
    constexpr int PE = 8;
    constexpr int HDP = GM::HDP, NKS = HDP / 32, ND = HD / 16;
    constexpr int PKP = GM::PKP;                        // K row pitch (160 B for hd 80: no pad); the last k-step of hd 80 reads 16 columns of the NEXT row
                                                        // (finite: the image has rows past WKEYS) against zero Q fragments
    constexpr int PVF = WKEYS + 4;                      // as in attnw_kernel
    constexpr int PE1 = GM::PE1;
    constexpr int PT = GM::PT;                          // table row pitch (224 B for hd 80)

    extern __shared__ __attribute__((aligned(16))) unsigned char smemw[];
    half_t* Ks0 = reinterpret_cast<half_t*>(smemw);     // 2 x [KROWS][PKP]
    half_t* Vts = Ks0 + 2 * GM::KIMG;                   // [HD][PVF]
    half_t* Es = Vts + HD * PVF;                        // [208][PE1]
    half_t* Wr = Es + WKEYS * PE1;                      // [8 waves][2 query blocks][16][PWR]: the kw terms of a query block, skewed (below)
    half_t* Ts = Wr + (PNT / 64) * 32 * GM::PWR;        // [64][PT]
    const unsigned ldsK = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smemw;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g_ = lane >> 4, li_ = lane & 15;
    const int nitems = p.S * p.heads;
    const half_t* __restrict__ Qb = reinterpret_cast<const half_t*>(p.Q);
    const half_t* __restrict__ Kb = reinterpret_cast<const half_t*>(p.K);
    const half_t* __restrict__ Vb = reinterpret_cast<const half_t*>(p.Vt);
    half_t* __restrict__ out = reinterpret_cast<half_t*>(p.out);

    // ---- once per workgroup: zero the V^T image (pad columns and keys past nk stay zero for good), stage E and the tables
    if (!VRM) for (int i = tid; i < HD * PVF / 4; i += PNT) reinterpret_cast<unsigned long long*>(Vts)[i] = 0ull;
    if (BIAS) {
        const half_t* __restrict__ prepE = reinterpret_cast<const half_t*>(p.win_prep);
        const half_t* __restrict__ prepT = prepE + WKEYS * 32;
        for (int i = tid; i < WKEYS * 4; i += PNT) store_piece(Es + (i >> 2) * PE1 + (i & 3) * PE, load_piece(prepE + i * PE));
        for (int i = tid; i < 64 * (HDP / PE); i += PNT) {
            const int r = i / (HDP / PE), c = i - r * (HDP / PE);
            store_piece(Ts + r * PT + c * PE, load_piece(prepT + r * HDP + c * PE));
        }
    }

    // ---- K: DMA piece j of the LDS image <- row min(j / LPPR, nk-1), piece min(j % LPPR, KPPR-1) (clamped: finite data;
    // keys past nk are masked in the softmax, the pad piece only meets zero Q fragments)
    constexpr int DPW = (GM::NDMA + 7) / 8;             // DMA instructions per wave
    unsigned kvoff[DPW];
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
        const int j = (wave + 8 * t) * 64 + lane;
        int r = j / GM::LPPR, c = j - r * GM::LPPR;
        r = r < p.nk ? r : p.nk - 1;
        c = c < GM::KPPR ? c : GM::KPPR - 1;
        kvoff[t] = (unsigned)(r * HD + c * PE) * 2u;
    }
    auto dma_k = [&](int sh, int buf) {
        const unsigned char* base = uniform_ptr_w(Kb + (long)sh * p.L * HD);
#pragma unroll
        for (int t = 0; t < DPW; ++t) {
            const int u = wave + 8 * t;
            if (u < GM::NDMA) {                          // wave-uniform
                const unsigned dst = __builtin_amdgcn_readfirstlane(ldsK + (unsigned)buf * (GM::KIMG * 2) + (unsigned)u * 1024u);
                AW_DMA(kvoff[t], base, dst);
            }
        }
    };

    // VRM: the V image = the item's WKEYS rows of HD halves, linear; piece j <- source piece min(j, nk * KPPR - 1) (clamped: rows past
    // nk hold finite data that only meets P = 0)
    constexpr int NDMAV = (WKEYS * GM::KPPR + 63) / 64;  // 33 for hd 80
    constexpr int DPWV = (NDMAV + 7) / 8;
    unsigned vvoff[VRM == 2 ? DPWV : 1];
    if constexpr (VRM == 2) {
#pragma unroll
        for (int t = 0; t < DPWV; ++t) {
            int j = (wave + 8 * t) * 64 + lane;
            j = j < p.nk * GM::KPPR ? j : p.nk * GM::KPPR - 1;
            vvoff[t] = (unsigned)j * 16u;
        }
    }
    auto dma_v = [&](int sh) {
        if constexpr (VRM == 2) {
            const unsigned char* base = uniform_ptr_w(Vb + (long)sh * p.L * HD);
#pragma unroll
            for (int t = 0; t < DPWV; ++t) {
                const int u = wave + 8 * t;
                if (u < NDMAV) {                         // wave-uniform
                    const unsigned dst = __builtin_amdgcn_readfirstlane(ldsK + (unsigned)(2 * GM::KIMG) * 2u + (unsigned)u * 1024u);
                    AW_DMA(vvoff[t], base, dst);
                }
            }
        }
    };
    constexpr int VPPR = WKEYS / PE;                    // 26 pieces per V^T row
    constexpr int VNR = (WKEYS * GM::KPPR + PNT - 1) / PNT;          // VRM = 1: pieces of the linear image per thread (5)
    constexpr int VN = VRM == 2 ? 1 : (VRM == 1 ? VNR : (HD * VPPR + PNT - 1) / PNT);
    Piece vreg[VN];
    half8_t qf[2][NKS];
    // Query blocks are WINDOW ROWS (round 5): block qb of wave w holds the KW <= 16 queries of grid row qy = 2 w + qb (lane column li = qx;
    // columns >= KW idle).  14 x 14 windows: 14 blocks of 14 on 7 waves — the same two blocks per wave as 13 blocks of 16 — and qy is
    // wave-uniform, qx the lane's own column: no per-lane divisions anywhere, and the kh term of the rel-pos bias needs no scatter (below).
    const bool wave_active = 2 * wave < p.KH;

    auto load_v = [&](int sh) {
        if constexpr (VRM == 2) return;
        if constexpr (VRM == 1) {      // piece i of the image <- source piece min(i, nk * KPPR - 1): rows past nk hold finite data that meets P = 0
            const half_t* __restrict__ Vg = Vb + (long)sh * p.L * HD;
#pragma unroll
            for (int u = 0; u < VN; ++u) {
                int i = tid + u * PNT;
                i = i < p.nk * GM::KPPR ? i : p.nk * GM::KPPR - 1;
                vreg[u] = load_piece(Vg + (long)i * PE);
            }
            return;
        }
        const half_t* __restrict__ Vg = Vb + (long)sh * HD * p.Lp;
#pragma unroll
        for (int u = 0; u < VN; ++u) {
            const int i = tid + u * PNT;
            const int d = i / VPPR, c = i - d * VPPR;
            vreg[u] = (i < HD * VPPR && c * PE < p.nk) ? load_piece(Vg + (long)d * p.Lp + c * PE) : zero_piece();
        }
    };
    auto store_v = [&]() {
        if constexpr (VRM == 2) return;
        if constexpr (VRM == 1) {
#pragma unroll
            for (int u = 0; u < VN; ++u) {
                const int i = tid + u * PNT;
                if (i < WKEYS * GM::KPPR) store_piece(Vts + i * PE, vreg[u]);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < VN; ++u) {
            const int i = tid + u * PNT;
            const int d = i / VPPR, c = i - d * VPPR;
            const int key0 = c * PE;
            if (i < HD * VPPR && key0 < p.nk) {
                Piece v = vreg[u];
                if (key0 + PE > p.nk) {                 // elements past the last key meet P = 0 but must not be NaN
                    half_t* e = reinterpret_cast<half_t*>(&v);
#pragma unroll
                    for (int j = 0; j < PE; ++j) if (key0 + j >= p.nk) e[j] = (half_t)0.f;
                }
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(Vts + d * PVF + c * PE);   // two 8-byte stores
                dst[0] = ((unsigned long long)v.w[1] << 32) | v.w[0];
                dst[1] = ((unsigned long long)v.w[3] << 32) | v.w[2];
            }
        }
    };
    auto load_q = [&](int sh) {
        const half_t* __restrict__ Qg = Qb + (long)sh * p.L * HD;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qy = 2 * wave + qb;
            const int row = qy * p.KW + li_;
            const bool okq = qy < p.KH && li_ < p.KW;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int d0 = ks * 32 + g_ * 8;
                qf[qb][ks] = (okq && d0 < HD) ? *reinterpret_cast<const half8_t*>(Qg + (long)row * HD + d0) : (half8_t)(0);
            }
        }
    };

    const float c1 = p.scale * W_LOG2E;
    int it = blockIdx.x;
    int buf = 0;
    bool need_q = false;
    if (it < nitems) { if (!(DBG & 16)) dma_k(it, 0); load_v(it); load_q(it); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // zero fill / E / tables / K image 0 complete

    for (; it < nitems; it += gridDim.x, buf ^= 1) {
        const half_t* Ks = Ks0 + buf * GM::KIMG;
        const int nxt = it + gridDim.x;
        if constexpr (VRM == 2) {
            // (every wave is past the barrier that ended the previous item: the V image and K image buf ^ 1 are free)
            if (!(DBG & 16)) {
                dma_v(it);                              // lands during the relcat + S^T phase; published by the barrier behind S^T
                if (nxt < nitems) dma_k(nxt, buf ^ 1);
            }
        } else {
            store_v();
            __syncthreads();                            // V^T of this item visible (its K image was fenced by the previous barrier)
            if (nxt < nitems) { dma_k(nxt, buf ^ 1); load_v(nxt); }   // in flight during this item's compute
        }

        // Loop-invariant per-lane predicates and addresses (key < nk, the rel-pos scatter conditions, ...) would be hoisted
        // out of the item loop and cost > 100 SGPRs / VGPRs of live state: re-derive them per item from opaque copies.
        int nk = p.nk, li = li_, g = g_;
        asm volatile("" : "+s"(nk));
        asm volatile("" : "+v"(li), "+v"(g));
        // Window mode: the grid rows of this window that lie past the image (64-wide grid, 14-wide windows: rows 8 .. 13 of the bottom window row)
        // are padding — window_unpartition drops their outputs (image_encoder.py:297-318).  Query blocks are window rows, so a wave whose two rows
        // are padding has nothing to compute for this item: it keeps its share of the DMA and the barriers and leaves the SIMD to its partner
        // (-2.8 % of the kernel, profiles/r06_e_attnwp_rowskip_ab.txt; static wave priorities and publishing V behind the softmax: +-1 %, same record).
        if (need_q) { load_q(it); need_q = false; }
        bool item_active = wave_active;
        if (p.win > 0) {
            const int w = (it / p.heads) % (p.nwx * p.nwy);
            item_active = wave_active && 2 * wave < p.gh - (w / p.nwx) * p.win;
        }
        if (item_active) {
            // ---- rel-pos operands of this wave's two query blocks: bf[qb] = the B fragment of the bias step S^T += E . bf, lane (g, li) =
            // slots 8g .. 8g+7 of query li.  Slot 8g + r carries the kh term for kh = KH-1 - (4g + r), slot 8g + 4 + r the kw term for
            // kw = KW-1 - (4g + r) (E, prepared per layer, has its ones accordingly: attnw_prep_kernel, layout 1).
            //   kh term: rel_h[q][kh] = q . Rh[qy - kh + KH-1] = q . Rh[qy + c], c = KH-1 - kh.  qy is the block's, so ONE MFMA chain against
            //     table rows qy .. qy+15 leaves exactly c = 4g + r in accumulator register r of lane (g, li): the slots are filled from the
            //     lane's own registers — no LDS, no scatter.
            //   kw term: rel_w[q][kw] = q . Rw[qx - kw + KW-1]: the table row depends on the lane's own column, a Toeplitz band.  The wave
            //     computes G[D][q] = q . Rw[D], D = 0 .. 31, and every lane stores its eight values UNCONDITIONALLY at the skewed position
            //     D - qx + 16 of its query's row (48 halves): position 16 + c then holds kw = KW-1 - c for c = 0 .. KW-1, out-of-band
            //     values land outside [16, 16 + KW) — inside the row, finite, and E is zero there.  Every position a lane reads back
            //     (16 + 4g + r) is rewritten for every item, so the area needs no initialisation.
            half8_t bf[2];
            if (DBG & 1) { bf[0] = (half8_t)(0); bf[1] = (half8_t)(0); }
            if (BIAS && !(DBG & 1)) {
                const float inv_scale = 1.0f / p.scale;
                half_t* wrows = Wr + (wave * 32) * GM::PWR;
                half8_t tw[2][NKS];
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks)
                        tw[jb][ks] = *reinterpret_cast<const half8_t*>(Ts + (32 + jb * 16 + li) * PT + ks * 32 + g * 8);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int qy = 2 * wave + qb;               // (< 16: table rows qy + li < 32, the rows past 2 KH - 2 are zero)
                    f32x4 ah = (f32x4)(0.f), aw0 = (f32x4)(0.f), aw1 = (f32x4)(0.f);
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        const half8_t th = *reinterpret_cast<const half8_t*>(Ts + (qy + li) * PT + ks * 32 + g * 8);
                        ah = __builtin_amdgcn_mfma_f32_16x16x32_f16(th, qf[qb][ks], ah, 0, 0, 0);
                        aw0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(tw[0][ks], qf[qb][ks], aw0, 0, 0, 0);
                        aw1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(tw[1][ks], qf[qb][ks], aw1, 0, 0, 0);
                    }
                    half_t* wr = wrows + (qb * 16 + li) * GM::PWR + (16 + 4 * g - li);       // position of D = 4g: D - qx + 16
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        wr[r] = (half_t)(aw0[r] * inv_scale);
                        wr[16 + r] = (half_t)(aw1[r] * inv_scale);
                        bf[qb][r] = (half_t)(ah[r] * inv_scale);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const half4_t w4 = *reinterpret_cast<const half4_t*>(wrows + (qb * 16 + li) * GM::PWR + 16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) bf[qb][4 + r] = w4[r];
                }
            }

            // ---- S^T for all 13 key blocks (rows past nk hold finite clamped data and are masked below), k-step outermost,
            // then the rel-pos contraction against E.  Fragment reads run 4 steps ahead of their MFMAs through a register
            // ring (inline-asm ds_read + counted lgkmcnt: left to itself the compiler issues each read right before its use).
            f32x4 s[2][WNKB];
#pragma unroll
            for (int kg = 0; kg < WNKB; ++kg) { s[0][kg] = (f32x4)(0.f); s[1][kg] = (f32x4)(0.f); }
            half8_t ring[4];
            if constexpr ((DBG & 2) != 0) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qb][0][r] = (float)bf[qb][r] + (float)qf[qb][0][r];      // (keeps the rel-pos phase and the Q loads alive)
            } else {
                const unsigned kbase = ldsK + (unsigned)buf * (GM::KIMG * 2) + (unsigned)(li * PKP + g * 8) * 2u;
                const unsigned ebase = ldsK + (unsigned)(2 * GM::KIMG + HD * PVF) * 2u + (unsigned)(li * PE1 + g * 8) * 2u;
                constexpr int TK = NKS * WNKB, TS = TK + (BIAS ? WNKB : 0);
                auto rd = [&](auto tc) {                    // fragment of step t: K block kg, k-step ks  |  E block t - TK
                    constexpr int t = decltype(tc)::value;
                    half8_t& dst = ring[t & 3];             // (named here: asm operands alone do not capture)
                    const unsigned kb = kbase, eb = ebase;
                    if constexpr (t < TK) { constexpr int ks = t / WNKB, kg = t - ks * WNKB; AW_DSR128(dst, kb, (kg * 16 * PKP + ks * 32) * 2); }
                    else AW_DSR128(dst, eb, (t - TK) * 16 * PE1 * 2);
                };
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                static_for<4>(rd);
                static_for<TS>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int left = TS - 1 - t;        // reads issued after this step's
                    if constexpr (left >= 3) AW_WAIT(3); else if constexpr (left == 2) AW_WAIT(2); else if constexpr (left == 1) AW_WAIT(1); else AW_WAIT(0);
                    if constexpr (t < TK) {
                        constexpr int ks = t / WNKB, kg = t - ks * WNKB;
                        s[0][kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[t & 3], qf[0][ks], s[0][kg], 0, 0, 0);
                        s[1][kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[t & 3], qf[1][ks], s[1][kg], 0, 0, 0);
                    } else {
                        constexpr int kg = t - TK;
                        s[0][kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[t & 3], bf[0], s[0][kg], 0, 0, 0);
                        s[1][kg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[t & 3], bf[1], s[1][kg], 0, 0, 0);
                    }
                    if constexpr (t + 4 < TS) rd(std::integral_constant<int, t + 4>{});
                    __builtin_amdgcn_sched_barrier(0);      // keep the MFMAs between their wait and the ring refill
                });
            }
            if constexpr (VRM == 2) {                   // this wave's share of V(it) (and of K(next)) has landed; then everybody's
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (nxt < nitems) load_q(nxt);              // Q of the next item: in flight during the softmax and PV

            // ---- one-pass softmax in the log2 domain, P^T fragments, O^T = V^T . P^T
            f32x4 o[2][ND];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int n = 0; n < ND; ++n) o[qb][n] = (f32x4)(0.f);
            float inv_l[2];
            half8_t pf[2][7];
            if constexpr ((DBG & 8) != 0) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    float a = 0.f;
#pragma unroll
                    for (int kg = 0; kg < WNKB; ++kg) a += s[qb][kg][0] + s[qb][kg][1] + s[qb][kg][2] + s[qb][kg][3];     // (keeps S^T alive)
                    inv_l[qb] = 1.0f;
#pragma unroll
                    for (int m = 0; m < 7; ++m) pf[qb][m] = (half8_t)((half_t)a);
                }
            } else
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                // nk > 192 on this path: only the last key block has keys to mask.  p = 2^(s c1 - max c1) as ONE packed fma per
                // pair (the maximum is taken over the raw scores, c1 > 0)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((WNKB - 1) * 16 + g * 4 + r >= nk) s[qb][WNKB - 1][r] = -INFINITY;
                float mx = -INFINITY;
#pragma unroll
                for (int kg = 0; kg < WNKB; ++kg)           // max(a, b) = med3(a, b, +inf): no IEEE-mode canonicalisation of the MFMA results
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = __builtin_amdgcn_fmed3f(mx, s[qb][kg][r], INFINITY);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const f32x2 c2 = {c1, c1};
                const f32x2 m2 = {-mx * c1, -mx * c1};
                f32x2 sum2 = {0.f, 0.f};
#pragma unroll
                for (int kg = 0; kg < 14; ++kg) {
                    if (kg < WNKB) {
                        const int kq = kg < WNKB ? kg : 0;
                        const f32x2 a0 = {s[qb][kq][0], s[qb][kq][1]}, a1 = {s[qb][kq][2], s[qb][kq][3]};
                        const f32x2 e0 = __builtin_elementwise_fma(a0, c2, m2), e1 = __builtin_elementwise_fma(a1, c2, m2);
                        f32x2 p0, p1;
                        p0[0] = __builtin_amdgcn_exp2f(e0[0]); p0[1] = __builtin_amdgcn_exp2f(e0[1]);
                        p1[0] = __builtin_amdgcn_exp2f(e1[0]); p1[1] = __builtin_amdgcn_exp2f(e1[1]);
                        sum2 += p0; sum2 += p1;
                        // contraction slot (m, g*8 + j): j < 4 -> key block 2m, reg j ; j >= 4 -> key block 2m+1, reg j-4
                        pf[qb][kg >> 1][(kg & 1) * 4 + 0] = (half_t)p0[0]; pf[qb][kg >> 1][(kg & 1) * 4 + 1] = (half_t)p0[1];
                        pf[qb][kg >> 1][(kg & 1) * 4 + 2] = (half_t)p1[0]; pf[qb][kg >> 1][(kg & 1) * 4 + 3] = (half_t)p1[1];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pf[qb][kg >> 1][(kg & 1) * 4 + r] = (half_t)0.f;
                    }
                }
                float sum = sum2[0] + sum2[1];
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                inv_l[qb] = 1.0f / sum;
            }
            if constexpr ((DBG & 4) != 0) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int m = 0; m < 7; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[qb][0][r] += (float)pf[qb][m][r] + (float)pf[qb][m][4 + r];           // (keeps the softmax alive)
            } else {
                // V^T fragments: key blocks 2m and 2m+1 of row n*16 + li (two 8-byte halves, 32 B apart); block 13 does not
                // exist (its P is 0): block 12 is read again instead.  Same 4-deep ring as above.
                constexpr int TP = 7 * ND;
                if constexpr (VRM) {
                    // row-major image: lane (g, i) addresses key 4g + i/4 of the block, elements n*16 + 4 (i%4) ..; the transposing read
                    // hands lane (g, li) V[4g + r][n*16 + li].  Compiler-visible loads (counted waits by the compiler), issued four steps
                    // ahead through the same ring; sched_barrier keeps them there.
                    const half_t* vbase = Vts + (g * 4 + (li >> 2)) * HD + (li & 3) * 4;
                    auto rdv = [&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        constexpr int m = t / ND, n = t - m * ND;
                        ring[t & 3] = w_frag_tr_2x4(vbase + (2 * m) * 16 * HD + n * 16, vbase + (m == 6 ? 2 * m : 2 * m + 1) * 16 * HD + n * 16);
                    };
                    static_for<4>(rdv);
                    static_for<TP>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        constexpr int m = t / ND, n = t - m * ND;
                        o[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[t & 3], pf[0][m], o[0][n], 0, 0, 0);
                        o[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[t & 3], pf[1][m], o[1][n], 0, 0, 0);
                        if constexpr (t + 4 < TP) rdv(std::integral_constant<int, t + 4>{});
                        __builtin_amdgcn_sched_barrier(0);
                    });
                } else {
                unsigned vb[ND];
#pragma unroll
                for (int n = 0; n < ND; ++n) vb[n] = ldsK + (unsigned)(2 * GM::KIMG) * 2u + (unsigned)((n * 16 + li) * PVF + g * 4) * 2u;
                auto rd = [&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int m = t / ND, n = t - m * ND;
                    half8_t& dst = ring[t & 3];
                    const unsigned va = vb[n];
                    AW_DSR2(dst, va, m * 8, m == 6 ? m * 8 : m * 8 + 4);
                };
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                static_for<4>(rd);
                static_for<TP>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int m = t / ND, n = t - m * ND;
                    constexpr int left = TP - 1 - t;
                    if constexpr (left >= 3) AW_WAIT(3); else if constexpr (left == 2) AW_WAIT(2); else if constexpr (left == 1) AW_WAIT(1); else AW_WAIT(0);
                    o[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[t & 3], pf[0][m], o[0][n], 0, 0, 0);
                    o[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[t & 3], pf[1][m], o[1][n], 0, 0, 0);
                    if constexpr (t + 4 < TP) rd(std::integral_constant<int, t + 4>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
                }
            }

            // ---- normalise and store; lane holds O[query li of qb][d = n*16 + g*4 + r].  This wave's prefetches (K DMA,
            // V^T, Q of the next item) are retired first: nothing ever waits for the stores below
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int s_idx = it / p.heads, h = it - s_idx * p.heads;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const int qy = 2 * wave + qb;                // the block's grid row (wave-uniform), li = the lane's grid column
                if (qy >= p.KH || li >= p.KW) continue;
                long row;
                if (p.win > 0) {                            // window mode: KH = KW = win
                    const int nw = p.nwx * p.nwy;
                    const int b = s_idx / nw, w = s_idx - b * nw;
                    const int wy = w / p.nwx, wx = w - wy * p.nwx;
                    const int gy = wy * p.win + qy, gx = wx * p.win + li;
                    if (gy >= p.gh || gx >= p.gw) continue;
                    row = (long)b * p.ntok + gy * p.gw + gx;
                } else {
                    row = (long)s_idx * p.ntok + qy * p.KW + li;
                }
                if constexpr (HD == 80) {
                    if (p.out8) { attn_store_mx8(p, o[qb], inv_l[qb], row, h, g); continue; }   // fp8 engine: MX-fp8 rows for the proj GEMM
                }
                if constexpr ((DBG & 32) != 0) { if (o[qb][0][0] * inv_l[qb] != 12345.f) continue; }
#pragma unroll
                for (int n = 0; n < ND; ++n) {
                    half4_t v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[qb][n][r] * inv_l[qb]);
                    *reinterpret_cast<half4_t*>(out + row * p.D + h * HD + n * 16 + g * 4) = v;
                }
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (VRM == 2) __builtin_amdgcn_s_barrier();   // the barrier behind the active waves' S^T phase
            need_q = wave_active;                       // (a wave that sat this item out has not asked for the next item's Q)
        }
        __syncthreads();                                // every wave is done with this item's K / V^T; the next K image has landed
    }
}


// E and the fp16 tables, once per layer (they are the same for every workgroup of the layer)
template <int HD>
// layout 0 (attnw_kernel): E[key][kh | KH + kw];  layout 1 (attnwp_kernel, query blocks = window rows): slot 8g + r <-> kh = KH-1 - (4g + r),
// slot 8g + 4 + r <-> kw = KW-1 - (4g + r) — the order in which that kernel's lanes hold the two terms (see there)
__global__ void attnw_prep_kernel(const float* __restrict__ tab_h, const float* __restrict__ tab_w, int KH, int KW, int nk,
                                  half_t* __restrict__ prep, int layout) {
    constexpr int HDP = (HD + 31) / 32 * 32;
    const int tid = threadIdx.x + blockIdx.x * blockDim.x, nth = blockDim.x * gridDim.x;
    for (int i = tid; i < WKEYS * 32; i += nth) {
        const int r = i >> 5, col = i & 31;
        const int kh = r / KW, kw = r - kh * KW;
        bool one;
        if (layout == 0) one = col == kh || col == KH + kw;
        else {
            const int c = 4 * (col >> 3) + (col & 3);
            one = (col & 4) ? (kw == KW - 1 - c) : (kh == KH - 1 - c);
        }
        prep[i] = (half_t)((r < nk && one) ? 1.f : 0.f);
    }
    half_t* T = prep + WKEYS * 32;
    for (int i = tid; i < 2 * 32 * HDP; i += nth) {
        const int tbl = i / (32 * HDP), rem = i - tbl * 32 * HDP;
        const int row = rem / HDP, d = rem - row * HDP;
        const int nj = 2 * (tbl ? KW : KH) - 1;
        T[i] = (half_t)((row < nj && d < HD) ? (tbl ? tab_w : tab_h)[(long)row * HD + d] : 0.f);
    }
}

template <int HD, int BIAS>
int launch_attnw_impl(const AttnParams& p, hipStream_t stream) {
    constexpr int HDP = (HD + 31) / 32 * 32;
    size_t lds = (size_t)(WKT * lds_pitch<half_t>(HDP) + HD * (WKEYS + 4) + 8) * sizeof(half_t);   // (+8: keeps Es 16-byte aligned)
    if (BIAS) lds += (size_t)(WKEYS + WQT) * 40 * sizeof(half_t);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attnw_kernel<HD, BIAS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    static const int persistent = cva_env_int("CVA_ATTNW_P", 1);
    if (BIAS) hipLaunchKernelGGL((attnw_prep_kernel<HD>), dim3(8), dim3(256), 0, stream, p.tab_h, p.tab_w, p.KH, p.KW, p.nk,
                                 reinterpret_cast<half_t*>(p.win_prep), (persistent && p.nk > (WNKB - 1) * 16) ? 1 : 0);
    // (no batch term in this test: the layouts a geometry fixes — row-major V, MX-fp8 rows — are decided once for max_batch and must hold
    //  for every smaller batch of the same geometry; with fewer items than CUs the persistent grid is simply smaller)
    if (BIAS && persistent && p.nk > (WNKB - 1) * 16) {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
                n_cu = 256;
        }
        static bool attr_p = false;
        if (!attr_p) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attnwp_kernel<HD, BIAS>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_p = true;
        }
        const size_t ldsp = AttnwpGeom<HD>::lds_bytes();
        const int items = p.S * p.heads;
        if (p.v_rm) {
            if constexpr (HD == 80 && BIAS) {
                static bool attr_v = false;
                if (!attr_v) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attnwp_kernel<HD, BIAS, 2>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e == hipSuccess && CVA_ABLATION_BUILD)
                        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attnwp_kernel<HD, BIAS, CVA_ABLATION_BUILD ? 1 : 2>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e != hipSuccess) return (int)e;
                    attr_v = true;
                }
#ifdef CVA_ABLATION
                static const int wdbg = cva_env_int("CVA_ATTNWP_DBG", 0);
                if (wdbg) {
                    bool hit = false;
#define CVA_AWP_DBG(D) if (wdbg == D) { hit = true; (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attnwp_kernel<HD, BIAS, 2, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                        hipLaunchKernelGGL((attnwp_kernel<HD, BIAS, 2, D>), dim3(items < n_cu ? items : n_cu), dim3(PNT), ldsp, stream, p); }
                    CVA_AWP_DBG(1) CVA_AWP_DBG(2) CVA_AWP_DBG(4) CVA_AWP_DBG(6) CVA_AWP_DBG(8) CVA_AWP_DBG(14) CVA_AWP_DBG(15) CVA_AWP_DBG(16) CVA_AWP_DBG(31) CVA_AWP_DBG(32) CVA_AWP_DBG(63)
#undef CVA_AWP_DBG
                    return hit ? (int)hipGetLastError() : (int)hipErrorInvalidValue;
                }
#endif
                static const int vreg = cva_env_int("CVA_VRM_REG", 0);      // ablation builds: the register-prefetch form of the V image, for A/B
                if (CVA_ABLATION_BUILD && vreg)
                    hipLaunchKernelGGL((attnwp_kernel<HD, BIAS, CVA_ABLATION_BUILD ? 1 : 2>), dim3(items < n_cu ? items : n_cu), dim3(PNT), ldsp, stream, p);
                else
                    hipLaunchKernelGGL((attnwp_kernel<HD, BIAS, 2>), dim3(items < n_cu ? items : n_cu), dim3(PNT), ldsp, stream, p);
                return (int)hipGetLastError();
            }
            return (int)hipErrorInvalidValue;
        }
        hipLaunchKernelGGL((attnwp_kernel<HD, BIAS>), dim3(items < n_cu ? items : n_cu), dim3(PNT), ldsp, stream, p);
        return (int)hipGetLastError();
    }
    if (p.v_rm || p.out8) return (int)hipErrorInvalidValue;        // the other window kernels read V^T and write fp16
    dim3 grid(p.S * p.heads);                            // one workgroup per (sequence, head); <= 2 query passes inside
    hipLaunchKernelGGL((attnw_kernel<HD, BIAS>), grid, dim3(WNT), lds, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

// fp16 only; returns -1 when the geometry is not covered (caller uses attention2)
int launch_attention_win(const AttnParams& p_in, hipStream_t stream) {
    AttnParams p = p_in;
    { static const int dbg = cva_env_int("CVA_ATTNW_DBG", 0); p.dbg = dbg; }   // ablation builds only
    if (p.nk > WKEYS || p.nk != p.L || (p.Lp < WKEYS && !p.v_rm)) return -1;
    const bool bias = p.tab_h && p.tab_w;
    if (bias && (p.KH > 16 || p.KW > 16 || p.nk != p.KH * p.KW || !p.win_prep)) return -1;      // (attn_takes_vrm mirrors these tests)
    if (p.hd == 80) return bias ? launch_attnw_impl<80, 1>(p, stream) : launch_attnw_impl<80, 0>(p, stream);
    if (p.hd == 64) return bias ? launch_attnw_impl<64, 1>(p, stream) : launch_attnw_impl<64, 0>(p, stream);
    return -1;
}

// (attention.h) fp8 output: the kernels with the MX-fp8 epilogue are attnwp_kernel<80, 1, *> and attn2_kernel<half, 80, *>
bool attn_takes_out8(const AttnParams& p) {
    static const int persistent = cva_env_int("CVA_ATTNW_P", 1), variant = cva_env_int("CVA_ATTN", 3);
    if (p.hd != 80 || !persistent || variant != 3) return false;
    const bool bias = p.tab_h && p.tab_w;
    const bool short_seq = p.nk <= WKEYS && p.nk == p.L && p.Lp >= WKEYS &&
                           !(bias && (p.KH > 16 || p.KW > 16 || p.nk != p.KH * p.KW || !p.win_prep));
    if (short_seq) return bias && p.nk > (WNKB - 1) * 16;       // batch-independent (see launch_attnw_impl)
    if (!bias) return true;
    return (p.KW == 64 && p.KH <= 64 && p.nk == p.KH * p.KW) || p.KH + p.KW <= 64;
}

// (attention.h) the geometry test the dispatcher of cellvit_abi.hip makes, for the row-major V forms of the two production kernels
bool attn_takes_vrm(const AttnParams& p, size_t elem_size) {
    static const int persistent = cva_env_int("CVA_ATTNW_P", 1), variant = cva_env_int("CVA_ATTN", 3), off = cva_env_int("CVA_NO_VRM", 0);
    if (elem_size != 2 || p.hd != 80 || !persistent || variant != 3 || off) return false;
    const bool bias = p.tab_h && p.tab_w;
    // would launch_attention_win take this layer (short key sequences, window OR global)?  Then only its persistent kernel reads row-major V
    const bool short_seq = p.nk <= WKEYS && p.nk == p.L && p.Lp >= WKEYS &&
                           !(bias && (p.KH > 16 || p.KW > 16 || p.nk != p.KH * p.KW || !p.win_prep));
    if (short_seq) return bias && p.nk > (WNKB - 1) * 16;       // batch-independent (see launch_attnw_impl)
    // Everything else runs attn2_kernel, whose row-major form (ablation builds) is SLOWER: 10.06 against
    // 9.19 ms per global SAM-H launch of 64 tiles (profiles/r04_d_vrm_kernels.txt: two ds_read_b64_tr_b16 per fragment instead of one
    // ds_read2_b64), while the qkv projection of a global layer already writes V^T with 16-byte stores.  Global layers keep V^T.
    static const int glob = cva_env_int("CVA_VRM_GLOBAL", 0);      // ablation builds: 1 = row-major V for attn2_kernel as well
    if (!glob) return false;
    if (!bias) return true;
    return (p.KW == 64 && p.KH <= 64 && p.nk == p.KH * p.KW) || p.KH + p.KW <= 64;
}

}  // namespace cva
