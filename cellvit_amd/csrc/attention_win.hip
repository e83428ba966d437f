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
    constexpr int PE1 = 32 + 8;                         // E / relcat row pitch: 80 B (odd multiple of 16 B: conflict-free b128 reads)

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


// E and the fp16 tables, once per layer (they are the same for every workgroup of the layer)
template <int HD>
__global__ void attnw_prep_kernel(const float* __restrict__ tab_h, const float* __restrict__ tab_w, int KH, int KW, int nk,
                                  half_t* __restrict__ prep) {
    constexpr int HDP = (HD + 31) / 32 * 32;
    const int tid = threadIdx.x + blockIdx.x * blockDim.x, nth = blockDim.x * gridDim.x;
    for (int i = tid; i < WKEYS * 32; i += nth) {
        const int r = i >> 5, col = i & 31;
        const int kh = r / KW, kw = r - kh * KW;
        prep[i] = (half_t)((r < nk && (col == kh || col == KH + kw)) ? 1.f : 0.f);
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
    if (BIAS) hipLaunchKernelGGL((attnw_prep_kernel<HD>), dim3(8), dim3(256), 0, stream, p.tab_h, p.tab_w, p.KH, p.KW, p.nk,
                                 reinterpret_cast<half_t*>(p.win_prep));
    dim3 grid(p.S * p.heads);                            // one workgroup per (sequence, head); <= 2 query passes inside
    hipLaunchKernelGGL((attnw_kernel<HD, BIAS>), grid, dim3(WNT), lds, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

// fp16 only; returns -1 when the geometry is not covered (caller uses attention2)
int launch_attention_win(const AttnParams& p_in, hipStream_t stream) {
    AttnParams p = p_in;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("CVA_ATTNW_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    if (p.nk > WKEYS || p.nk != p.L || p.Lp < WKEYS) return -1;
    const bool bias = p.tab_h && p.tab_w;
    if (bias && (p.KH > 16 || p.KW > 16 || p.nk != p.KH * p.KW || !p.win_prep)) return -1;
    if (p.hd == 80) return bias ? launch_attnw_impl<80, 1>(p, stream) : launch_attnw_impl<80, 0>(p, stream);
    if (p.hd == 64) return bias ? launch_attnw_impl<64, 1>(p, stream) : launch_attnw_impl<64, 0>(p, stream);
    return -1;
}

}  // namespace cva
