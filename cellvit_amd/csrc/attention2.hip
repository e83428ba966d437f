// Attention v2 for gfx950: transposed flash attention with the decomposed relative-position bias
// fused in (no separate rel-pos kernel, no bias tables in HBM).
//
//   S^T = K · Q^T   (A = K tile rows, B = Q rows)   -> lane l holds keys g*4+r of ONE query (l&15):
//                                                     the softmax row reductions are lane-local + 2 shuffles
//   O^T = V^T · P^T (A = V^T rows,   B = P^T)       -> P^T is consumed straight from the S^T registers
//                                                     (the contraction index is re-labelled so that the
//                                                     C-fragment of S^T IS the B-fragment of P^T);
//                                                     accumulators keep the same lane<->query map, so the
//                                                     running max / sum / rescale never cross lanes.
// One workgroup = 4 waves x 32 queries (2 query blocks of 16) of one (sequence, head); K and V^T tiles
// of 64 keys are staged through LDS and shared by the 8 query blocks.
//
// Decomposed rel-pos (SAM/image_encoder.py:354-392): bias[q,(kh,kw)] = q·Rh[qy-kh+KH-1] + q·Rw[qx-kw+KW-1].
// At block start each wave computes G = Q·tab^T for its 32 queries with MFMAs and scatters the needed
// diagonal band into an LDS table relcat[q][0..KH) ++ [KH..KH+KW) (pre-divided by the softmax scale).
//   BIAS 1 (KH+KW <= 64: windows, small tiles): the bias is one or two extra contraction steps of the
//          S^T MFMA against a per-tile one-hot matrix E[key][kh | KH+kw] — no per-element lookups.
//   BIAS 2 (KW == 64 == key tile: 1024-px tiles, global blocks): kh is constant per key tile (one LDS
//          scalar per query and tile), the kw term is tile-invariant and lives in registers.
//
// Round 5 — the softmax's arithmetic moved INTO the contraction (the kernel is VALU-issue bound: 6 vector instructions per score next to
// 44 MFMAs per wave and key tile, profiles/r03_s_kernel_insts.txt):
//   * Q is pre-scaled by scale * log2(e) when its fragments are loaded (the reference rounds q * scale to fp16 as well,
//     image_encoder.py:244 under autocast), the rel-pos tables' products follow (relcat holds bias * log2 e): S^T leaves the MFMAs in
//     the log2 domain — no per-score multiply;
//   * the accumulators of S^T are not zeroed but INITIALISED with what used to be added per score afterwards: the tile-invariant kw
//     bias (BIAS 2) and the per-tile shift  bh(kt) - m_ref  (the kh bias of the tile minus the lazy reference maximum);
//   * fp16, hd 80 (SAM-H, the production case): the shift does not even cost the initialising add — the head dim is zero-padded from
//     80 to 96 contraction slots, slots 80 / 81 carry  1.0  on the K side (set once in the LDS tile's pad columns) and the shift as an
//     fp16 hi + lo pair on the Q side (patched per tile: 2 conversions per query block), so C = the kw registers as they are.
// Per score that leaves: maximum (v_max3), exp2, row-sum add, half a conversion.
#include "attention.h"

#include <type_traits>

namespace cva {

namespace {

constexpr int KT = 64, NT = 256, QW = 32, QT = 128;
constexpr float LOG2E = 1.4426950408889634f;

template <typename T> struct Pack4;   // 4 consecutive T elements
template <> struct Pack4<half_t> { typedef _Float16 type __attribute__((ext_vector_type(4))); };
template <> struct Pack4<float> { typedef float type __attribute__((ext_vector_type(4))); };

template <typename T>
__device__ __forceinline__ void set_frag(typename Traits<T>::Frag& f, int j, float v);
template <>
__device__ __forceinline__ void set_frag<half_t>(Traits<half_t>::Frag& f, int j, float v) { f.v[j] = (half_t)v; }
template <>
__device__ __forceinline__ void set_frag<float>(Traits<float>::Frag& f, int j, float v) { f.v[j] = v; }

template <typename T>
__device__ __forceinline__ typename Traits<T>::Frag frag_from_2x4(const T* p0, const T* p1) {
    typename Traits<T>::Frag f;
    const typename Pack4<T>::type a = *reinterpret_cast<const typename Pack4<T>::type*>(p0);
    const typename Pack4<T>::type b = *reinterpret_cast<const typename Pack4<T>::type*>(p1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
    return f;
}

// V ROW-major (VRM, fp16): the V tile is staged as it lies in memory, [key][HD] with the unpadded pitch of HD halves, and the PV operand
// (A = V^T rows d = n*16 + li, keys 4g .. 4g+3 of two key blocks) comes out of the transposing LDS read: lane (g, i) of a 16-lane group
// passes the address of key 4g + i/4, elements d0 + 4 (i%4) .. +3, and receives V[4g + r][d0 + i], r = 0..3 (measured:
// tools/probes/probe_ds_read_tr.hip, profiles/r04_probe_ds_read_tr.txt).  With HD = 80 the pitch is 40 dwords: the 8 keys a half-wave
// touches start on 8 distinct multiples of 8 banks.
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8_t frag_tr_2x4(const half_t* p0, const half_t* p1) {
    typedef __attribute__((address_space(3))) fp16x4_t* lp;
    const h4_t a = __builtin_bit_cast(h4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(__attribute__((address_space(3))) void*)const_cast<half_t*>(p0)));
    const h4_t b = __builtin_bit_cast(h4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(__attribute__((address_space(3))) void*)const_cast<half_t*>(p1)));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename T, int HD, int BIAS, int NBK, int VRM = 0>
__global__ __launch_bounds__(NT, 2) void attn2_kernel(const AttnParams p) {
    static_assert(!VRM || (sizeof(T) == 2 && HD == 80), "row-major V: fp16, hd 80 (conflict-free pitch of the transposing read)");
    using TR = Traits<T>;
    using Frag = typename TR::Frag;
    constexpr int PE = TR::PIECE;
    constexpr int HDP = (HD + 31) / 32 * 32, NKS = HDP / 32, ND = HD / 16;
    // V^T rows are read as 8-byte halves of a fragment, which the compiler merges into ds_read2_b64 (32-bank mapping):
    // a 136-byte pitch (34 dwords) puts the 16 rows of a lane group on 16 distinct bank pairs under both the 32- and the
    // 64-bank mapping (144 B collided pairwise: SQ_LDS_BANK_CONFLICT was 1.7x the kernel's active LDS cycles).  Rows are
    // then only 8-byte aligned, so the staging writes of V^T are two 8-byte stores per piece.
    constexpr int PK = lds_pitch<T>(HDP), PV = sizeof(T) == 2 ? KT + 4 : lds_pitch<T>(KT);
    constexpr int EW = NBK * 32;                       // one-hot width (BIAS 1)
    constexpr int PE1 = lds_pitch<T>(EW);              // pitch of E and relcat rows (BIAS 1)
    constexpr int RC2 = 128 + 8;                       // relcat row pitch (BIAS 2): KH + KW <= 128

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ks = reinterpret_cast<T*>(smem_raw);            // [KT][PK]
    T* Vts = Ks + KT * PK;                             // [HD][PV]
    T* Es = Vts + HD * PV;                             // BIAS 1: [KT][PE1]
    T* Rc = Es + (BIAS == 1 ? KT * PE1 : 0);           // relcat: BIAS 1 [QT][PE1], BIAS 2 [QT][RC2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    // XCD-aware remap: workgroups are dealt round-robin to the 8 XCDs in launch order; give every XCD a contiguous range
    // of (sequence, head) pairs so that all query blocks of one head share that XCD's L2 copy of K / V^T
    // (measured before: K / V fetched 8x from the fabric by the global blocks, FETCH_SIZE 1.37 GB vs 0.25 GB algorithmic)
    const int lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int qblk = lin % gridDim.x;
    const int q0 = qblk * QT + wave * QW;               // first query of this wave
    const int sh = lin / gridDim.x;

    const T* __restrict__ Qg = reinterpret_cast<const T*>(p.Q) + (long)sh * p.L * HD;
    const T* __restrict__ Kg = reinterpret_cast<const T*>(p.K) + (long)sh * p.L * HD;
    const T* __restrict__ Vg = reinterpret_cast<const T*>(p.Vt) + (VRM ? (long)sh * p.L * HD : (long)sh * HD * p.Lp);

    // SLOT: the per-tile softmax shift rides in contraction slots HD, HD + 1 of the zero-padded head dim (see the header)
    constexpr bool SLOT = sizeof(T) == 2 && HD == 80 && BIAS == 2;
    if (HDP > HD) {
        for (int i = tid; i < KT * (HDP - HD); i += NT) {
            const int r = i / (HDP - HD), c = i - r * (HDP - HD);
            Ks[r * PK + HD + c] = TR::from_float((SLOT && c < 2) ? 1.f : 0.f);
        }
    }
    const float c1 = p.scale * LOG2E;                  // Q is pre-scaled by it: scores and rel-pos terms come out of the MFMAs in the log2 domain

    // ---- Q fragments (B operand of S^T): lane -> query li of block qb, head-dim slice g*8.. ----
    Frag qf[2][NKS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int row = q0 + qb * 16 + li;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = ks * 32 + g * 8;
            qf[qb][ks] = (row < p.L && d0 < HD) ? TR::load_frag(Qg + (long)row * HD + d0) : TR::zero_frag();
#pragma unroll
            for (int j = 0; j < 8; ++j) set_frag<T>(qf[qb][ks], j, TR::to_float(qf[qb][ks].v[j]) * c1);
        }
    }

    // K / V^T tiles are fetched one tile ahead into registers (issue early, write to LDS after the barrier):
    // the global-memory latency of tile kt+1 hides behind the MFMAs of tile kt.
    constexpr int KPPR = HD / PE, VPPR = KT / PE;
    constexpr int KN = (KT * KPPR + NT - 1) / NT, VN = VRM ? KN : (HD * VPPR + NT - 1) / NT;
    Piece kreg[KN], vreg[VN];
    // LAST (compile time): only the last key tile can hold keys >= nk; every other tile skips the bounds tests and the
    // per-element tail zeroing (the optimiser turned those into ~140 selects per tile, executed on every tile)
    auto fetch = [&](int kt, auto lastc) {
        constexpr bool LAST = decltype(lastc)::value;
#pragma unroll
        for (int u = 0; u < KN; ++u) {
            const int i = tid + u * NT;
            const int r = i / KPPR, c = i - r * KPPR;
            const int key = kt * KT + r;
            kreg[u] = (i < KT * KPPR && (!LAST || key < p.nk)) ? load_piece(Kg + (long)key * HD + c * PE) : zero_piece();
        }
        if constexpr (VRM) {      // the V tile is one contiguous run of KT rows of HD halves, like the K tile
#pragma unroll
            for (int u = 0; u < VN; ++u) {
                const int i = tid + u * NT;
                const int r = i / KPPR, c = i - r * KPPR;
                const int key = kt * KT + r;
                vreg[u] = (i < KT * KPPR && (!LAST || key < p.nk)) ? load_piece(Vg + (long)key * HD + c * PE) : zero_piece();
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < VN; ++u) {
            const int i = tid + u * NT;
            const int d = i / VPPR, c = i - d * VPPR;
            const int key0 = kt * KT + c * PE;
            Piece v = zero_piece();
            if (i < HD * VPPR) {
                v = load_piece(Vg + (long)d * p.Lp + key0);
                if (LAST && key0 + PE > p.nk) {         // never let stale bytes past the last key meet P = 0
                    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
                    for (int j = 0; j < PE; ++j) if (key0 + j >= p.nk) e[j] = TR::from_float(0.f);
                }
            }
            vreg[u] = v;
        }
    };
    const int ntiles = (p.nk + KT - 1) / KT;
    const std::integral_constant<bool, true> LAST_T{};
    const std::integral_constant<bool, false> FULL_T{};
    if (ntiles == 1) fetch(0, LAST_T); else fetch(0, FULL_T);

    // ---- decomposed rel-pos: relcat[q][kh] = q·tab_h[qy-kh+KH-1] / scale ; relcat[q][KH+kw] likewise ----
    const float inv_scale = 1.0f / p.scale;
    if (BIAS != 0) {
        constexpr int RCP = (BIAS == 1) ? PE1 : RC2;
        T* myrc = Rc + (wave * QW) * RCP;
        if constexpr (sizeof(T) == 2 && (QW * RCP) % 8 == 0) {            // 16-byte stores (rows are 16-byte multiples)
            for (int i = lane; i < QW * RCP / 8; i += 64) store_piece(myrc + i * 8, zero_piece());
        } else {
            for (int i = lane; i < QW * RCP; i += 64) myrc[i] = TR::from_float(0.f);
        }
        // BIAS 1 (2*K - 1 <= 32 table rows): both tables are staged once per block, coalesced, into LDS as T-typed rows of
        // pitch PK (zero padded to 32 rows x HDP columns; aliased onto the K / V^T tile area, which is idle until the key
        // loop), so that the table fragments below are single 16-byte LDS reads instead of 8 scattered global loads.
        T* Ts = Ks;
        const bool STAGED = (BIAS == 1) && (2 * 32 * PK <= KT * PK + HD * PV) && (HD % 4 == 0) && p.KH <= 16 && p.KW <= 16;   // block-uniform
        if (STAGED) {
            for (int i = tid; i < 2 * 32 * (PK / PE); i += NT) store_piece(Ts + i * PE, zero_piece());
            __syncthreads();
            const int njh = 2 * p.KH - 1, njw = 2 * p.KW - 1;
            const int nq = (njh + njw) * (HD / 4);
            for (int i = tid; i < nq; i += NT) {
                const int row = i / (HD / 4), c4 = i - row * (HD / 4);
                const bool isw = row >= njh;
                const float* src = (isw ? p.tab_w + (long)(row - njh) * HD : p.tab_h + (long)row * HD) + c4 * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                T* dst = Ts + ((isw ? 32 + row - njh : row)) * PK + c4 * 4;
                dst[0] = TR::from_float(v[0]); dst[1] = TR::from_float(v[1]); dst[2] = TR::from_float(v[2]); dst[3] = TR::from_float(v[3]);
            }
            __syncthreads();
        }
#pragma unroll 1
        for (int tbl = 0; tbl < 2; ++tbl) {
            const float* __restrict__ tab = tbl == 0 ? p.tab_h : p.tab_w;
            const int Ksz = tbl == 0 ? p.KH : p.KW;
            const int off = tbl == 0 ? 0 : p.KH;
            const int nj = 2 * Ksz - 1;
#pragma unroll 1
            for (int jb = 0; jb * 16 < nj; ++jb) {
                if (BIAS == 2) {
                    // only the table rows j = c - kk + Ksz - 1 with 0 <= kk < Ksz are ever scattered: for the wave's 32 consecutive
                    // queries c spans [clo, chi] (one grid row: a single qy; 32 consecutive qx), i.e. rows [clo, chi + Ksz - 1] —
                    // 16-row blocks outside that band are skipped (wave-uniform: 5 of 8 blocks for the row table, 6 for the column one)
                    const int qlo = min(q0, p.L - 1), qhi = min(q0 + QW - 1, p.L - 1);
                    const int clo = tbl == 0 ? qlo / p.KW : qlo % p.KW, chi = tbl == 0 ? qhi / p.KW : (qlo / p.KW == qhi / p.KW ? qhi % p.KW : p.KW - 1);
                    const int cl2 = tbl == 0 ? clo : (qlo / p.KW == qhi / p.KW ? clo : 0);
                    if (jb * 16 + 15 < cl2 || jb * 16 > chi + Ksz - 1) continue;
                }
                Frag tf[NKS];
                const int j = jb * 16 + li;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const int d0 = ks * 32 + g * 8;
                    tf[ks] = TR::zero_frag();
                    if (STAGED) {
                        tf[ks] = TR::load_frag(Ts + (tbl * 32 + j) * PK + d0);
                    } else if (j < nj && d0 < HD) {
                        const float* src = tab + (long)j * HD + d0;      // (HD % 8 == 0: two 16-byte loads)
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(src), t1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { set_frag<T>(tf[ks], e, t0[e]); set_frag<T>(tf[ks], 4 + e, t1[e]); }
                    }
                }
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    f32x4 acc = (f32x4)(0.f);
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) TR::mma(tf[ks], qf[qb][ks], acc);
                    // acc[r] = q(li) · tab[jb*16 + g*4 + r]
                    const int q = q0 + qb * 16 + li;
                    if (q < p.L) {
                        const int qy = q / p.KW, qx = q - qy * p.KW;
                        const int c = tbl == 0 ? qy : qx;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int jj = jb * 16 + g * 4 + r;
                            const int kk = c - jj + Ksz - 1;          // image_encoder.py:347-351
                            if (jj < nj && kk >= 0 && kk < Ksz)
                                myrc[(qb * 16 + li) * RCP + off + kk] = TR::from_float(acc[r] * inv_scale);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // (relcat = (Q c1) . tab / scale = bias * log2 e: no further factor)
    // BIAS 1: Q-side bias fragments (rows of relcat);  BIAS 2: tile-invariant kw terms in registers
    Frag bf[2][NBK];
    float bw[2][4][4];
    if (BIAS == 1) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb2 = 0; kb2 < NBK; ++kb2)
                bf[qb][kb2] = TR::load_frag(Rc + (wave * QW + qb * 16 + li) * PE1 + kb2 * 32 + g * 8);
    } else if (BIAS == 2) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    bw[qb][kb][r] = TR::to_float(Rc[(wave * QW + qb * 16 + li) * RC2 + p.KH + kb * 16 + g * 4 + r]);
    }

    f32x4 o[2][ND];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int n = 0; n < ND; ++n) o[qb][n] = (f32x4)(0.f);
    // m_run: the lazy reference maximum (log2 domain).  It starts at 0 and the FIRST tile always takes the update path below, which then
    // sets it to the tile's row maximum (the accumulators are still zero: any finite factor is harmless)
    float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};

    const bool wave_active = q0 < p.L;                 // waves whose 32 queries are all padding only help staging

    auto step = [&](const int kt, auto lastc) {
        constexpr bool LAST = decltype(lastc)::value;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KN; ++u) {
            const int i = tid + u * NT;
            if (i < KT * KPPR) { const int r = i / KPPR, c = i - r * KPPR; store_piece(Ks + r * PK + c * PE, kreg[u]); }
        }
#pragma unroll
        for (int u = 0; u < VN; ++u) {
            const int i = tid + u * NT;
            if constexpr (VRM) {
                if (i < KT * KPPR) store_piece(Vts + i * PE, vreg[u]);      // [key][HD], pitch HD: the tile image is linear
            } else if (i < HD * VPPR) {
                const int d = i / VPPR, c = i - d * VPPR;
                if constexpr (sizeof(T) == 2) {
                    unsigned long long* dst = reinterpret_cast<unsigned long long*>(Vts + d * PV + c * PE);
                    dst[0] = ((unsigned long long)vreg[u].w[1] << 32) | vreg[u].w[0];
                    dst[1] = ((unsigned long long)vreg[u].w[3] << 32) | vreg[u].w[2];
                } else {
                    store_piece(Vts + d * PV + c * PE, vreg[u]);
                }
            }
        }
        if (BIAS == 1) {   // one-hot rows E[key][kh] = E[key][KH + kw] = 1
            constexpr int PPR = EW / PE;
            for (int i = tid; i < KT * PPR; i += NT) {
                const int r = i / PPR, c = i - r * PPR;
                const int key = kt * KT + r;
                const int kh = key / p.KW, kw = key - kh * p.KW;
                Piece v = zero_piece();
                if (key < p.nk) {
                    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
                    for (int j = 0; j < PE; ++j) {
                        const int col = c * PE + j;
                        if (col == kh || col == p.KH + kw) e[j] = TR::from_float(1.f);
                    }
                }
                store_piece(Es + r * PE1 + c * PE, v);
            }
        }
        __syncthreads();
        if (!LAST) { if (kt + 2 == ntiles) fetch(kt + 1, LAST_T); else fetch(kt + 1, FULL_T); }
        if (!wave_active) return;
        const int nkb = LAST ? min(4, (p.nk - kt * KT + 15) / 16) : 4;      // key blocks of this tile that hold real keys

        // ---- S^T blocks: s[qb][kb][r] = log2-domain exponent of (query li of qb, key kb*16 + g*4 + r) RELATIVE to the lazy reference
        // maximum:  (q c1) . k  +  kw bias  +  kh bias of the tile  -  m_run.   Everything but the dot product is the accumulators'
        // initial value (or, SLOT, two contraction slots of the padded head dim) — see the header.
        // (k-step outermost: the 8 MFMAs of a k-step write 8 different accumulators, so dependent MFMAs are 8 apart
        //  instead of 2 — the matrix pipe does not stall on its own result)
        float shift[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float bh = 0.f;
            if (BIAS == 2) bh = TR::to_float(Rc[(wave * QW + qb * 16 + li) * RC2 + kt]);   // kh == kt
            shift[qb] = bh - m_run[qb];
            if constexpr (SLOT) {      // slots 80 / 81 = elements 0 / 1 of the last k-step's fragment on the lanes with g == 2
                const half_t hi = (half_t)shift[qb];
                const half_t lo = (half_t)(shift[qb] - (float)hi);
                qf[qb][NKS - 1].v[0] = g == 2 ? hi : qf[qb][NKS - 1].v[0];
                qf[qb][NKS - 1].v[1] = g == 2 ? lo : qf[qb][NKS - 1].v[1];
            }
        }
        f32x4 s[2][4];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if constexpr (SLOT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qb][kb][r] = bw[qb][kb][r];
                } else if (BIAS == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qb][kb][r] = bw[qb][kb][r] + shift[qb];
                } else {
                    s[qb][kb] = (f32x4)(shift[qb]);
                }
            }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if (kb >= nkb) continue;                          // wave-uniform: masked to -inf below
                const Frag kf = TR::load_frag(Ks + (kb * 16 + li) * PK + ks * 32 + g * 8);
                TR::mma(kf, qf[0][ks], s[0][kb]);
                TR::mma(kf, qf[1][ks], s[1][kb]);
            }
        }
        if (BIAS == 1) {
#pragma unroll
            for (int kb2 = 0; kb2 < NBK; ++kb2) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb >= nkb) continue;
                    const Frag ef = TR::load_frag(Es + (kb * 16 + li) * PE1 + kb2 * 32 + g * 8);
                    TR::mma(ef, bf[0][kb2], s[0][kb]);
                    TR::mma(ef, bf[1][kb2], s[1][kb]);
                }
            }
        }
        // ---- online softmax per query, LAZY reference maximum (log2 domain) ----
        //   p = 2^t,  t = the accumulator (exponent relative to m_run).
        // m_run is only raised when some t of the tile exceeds LAZY_TAU (checked wave-wide with one ballot; always on the first tile,
        // which sets it); otherwise the tile needs NO cross-lane reduction and NO rescale of the output accumulators — p may then be as
        // large as 2^LAZY_TAU, harmless in fp16/fp32.  The row sum is kept as a per-lane partial (the four lanes of a query are combined
        // once, after the key loop).  Mathematically identical to the eager form.
        constexpr float LAZY_TAU = 8.0f;
        const bool ragged = LAST && (p.nk & (KT - 1)) != 0;                   // only the last tile can hold invalid keys
        Frag pf[2][2];
        float tmax[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (LAST && ragged) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int key = kt * KT + kb * 16 + g * 4 + r; s[qb][kb][r] = key < p.nk ? s[qb][kb][r] : -INFINITY; }
            }
            float mx = __builtin_fmaxf(__builtin_fmaxf(s[qb][0][0], s[qb][0][1]), s[qb][0][2]);      // (nested pairs: v_max3_f32)
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][0][3]), s[qb][1][0]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][1][1]), s[qb][1][2]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][1][3]), s[qb][2][0]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][2][1]), s[qb][2][2]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][2][3]), s[qb][3][0]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][3][1]), s[qb][3][2]);
            tmax[qb] = __builtin_fmaxf(mx, s[qb][3][3]);
        }
        if (kt == 0 || __any((tmax[0] > LAZY_TAU) || (tmax[1] > LAZY_TAU))) {      // wave-uniform, rare after the first tiles
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float mx = tmax[qb];
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                // first tile: the reference becomes the tile's row maximum whatever its sign (o and l are still zero);
                // later: it is only ever raised
                const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run[qb] += delta;
                l_run[qb] *= alpha;
#pragma unroll
                for (int n = 0; n < ND; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[qb][n][r] *= alpha;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qb][kb][r] -= delta;
            }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(s[qb][kb][r]);
                    rs += pv;
                    // contraction slot (m, g*8 + j): j < 4 -> key block 2m, reg j ; j >= 4 -> key block 2m+1, reg j-4
                    set_frag<T>(pf[qb][kb >> 1], (kb & 1) * 4 + r, pv);
                }
            l_run[qb] += rs;                                       // per-lane partial row sum
        }
        // ---- O^T += V^T P^T : A = V^T rows d = n*16 + li, keys (2m)*16 + g*4 + {0..3} and (2m+1)*16 + g*4 + {0..3} ----
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (2 * m >= nkb) continue;                           // all P of these 32 keys are exactly 0
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                Frag vf;
                if constexpr (VRM) {
                    const half_t* vkey = reinterpret_cast<const half_t*>(Vts) + ((2 * m) * 16 + g * 4 + (li >> 2)) * HD + n * 16 + (li & 3) * 4;
                    vf.v = frag_tr_2x4(vkey, vkey + 16 * HD);
                } else {
                    const T* vrow = Vts + (n * 16 + li) * PV + g * 4;
                    vf = frag_from_2x4<T>(vrow + (2 * m) * 16, vrow + (2 * m + 1) * 16);
                }
                TR::mma(vf, pf[0][m], o[0][n]);
                TR::mma(vf, pf[1][m], o[1][n]);
            }
        }
    };
    for (int kt = 0; kt + 1 < ntiles; ++kt) step(kt, FULL_T);
    step(ntiles - 1, LAST_T);

    // ---- normalise; lane holds O[query li of qb][d = n*16 + g*4 + r] ----
    const int s_idx = sh / p.heads, h = sh - s_idx * p.heads;
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float lsum = l_run[qb];                                // combine the four lanes' partial row sums (all lanes active)
        lsum += __shfl_xor(lsum, 16);
        lsum += __shfl_xor(lsum, 32);
        const int qg = q0 + qb * 16 + li;
        if (qg >= p.L) continue;
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
        const float inv = 1.0f / lsum;
        if constexpr (sizeof(T) == 2 && HD == 80) {
            if (p.out8) { attn_store_mx8(p, o[qb], inv, row, h, g); continue; }      // fp8 engine: MX-fp8 rows for the proj GEMM
        }
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            typename Pack4<T>::type v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = TR::from_float(o[qb][n][r] * inv);
            *reinterpret_cast<typename Pack4<T>::type*>(out + row * p.D + h * HD + n * 16 + g * 4) = v;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------------------
// attn2d_kernel (round 5): the global blocks of SAM-H at production geometry — fp16, hd 80, 64-wide key grids (BIAS 2), nk a multiple of 64 — with K and
// V^T tiles staged by LDS-DMA through a ring of THREE stages and ONE barrier per key tile.  attn2_kernel fetches a tile into 24 VGPRs one tile ahead, writes it
// to LDS between two barriers (nine LDS stores per thread and tile) and needs zero-padded K rows; here
//   * the K tile is the 64 x 80-half block as it lies in memory (10 one-KiB pieces, rows of 160 B: conflict-free for the fragment reads because the
//     two 16-byte columns a 16-lane group touches are 16 B apart); the padded contraction slots 80 .. 95 never exist in LDS: lanes g >= 2 of the last
//     k-step read a 32-byte constant block instead (slots 80 / 81 = 1.0 for the softmax shift, see attn2_kernel's header, the rest zero);
//   * the V^T tile is 80 rows of 128 B, 16-byte pieces XOR-swizzled by (row >> 1) & 7 at the SOURCE (the DMA's LDS image is lane-linear) so that the
//     8-byte fragment halves of 16 rows fall on distinct bank groups;
//   * tile t+2's twenty pieces are issued at the top of tile t; at the end of tile t a COUNTED wait retires tile t+1's (requested a whole tile earlier) and
//     leaves the newest in flight, then the one barrier publishes them.  Three stages + the kh relcat rows (the kw terms live in registers after the prologue,
//     their scratch is stage 2) are 78 KB: two workgroups per CU as before.
// Everything else (transposed flash attention, rel-pos bias from the accumulators' initial value, lazy reference maximum, epilogues) is attn2_kernel's.
//
// PIPE = 1 (round 5, second half): the same kernel software-pipelined inside the wave — the S^T MFMAs of key tile t + 1 are issued in the SAME basic block as
// the exponentials / conversions of tile t and the PV MFMAs of tile t, so that one wave alone keeps the matrix pipe fed while its vector instructions issue
// (the PIPE = 0 wave runs S^T -> maximum -> exp -> PV strictly in sequence and relies on the SIMD's second wave for all overlap).  K runs two tiles ahead of V:
// at the top of tile t the pieces of K(t + 3) and V(t + 2) are requested (K(j) and V(j) live in stage j % 3), the counted wait at the end of tile t leaves
// exactly that batch in flight, so K(t + 2) and V(t + 1) — requested a whole tile earlier — have landed behind the tile's one barrier.  The arithmetic and
// its order per score are PIPE = 0's: the outputs are bit-identical.
// DBG (ablation library only, CVA_ATTN2D_DBG; results are wrong by construction): 1 exponentials -> multiplies, 2 no S^T MFMAs, 4 no PV MFMAs, 8 no per-tile wait /
// barrier, 16 K fragments not re-read, 32 V^T fragments not re-read, 64 no softmax arithmetic at all, 128 no tile DMA, 256 one key tile only (prologue + epilogue).
template <int PIPE = 0, int DBG = 0>
__global__ __launch_bounds__(NT, 2) void attn2d_kernel(const AttnParams p) {
    using T = half_t;
    using TR = Traits<half_t>;
    using Frag = TR::Frag;
    constexpr int HD = 80, NKS = 3, ND = 5;
    constexpr int RCK = 64 + 8;                        // relcat row: the kh terms only (the kw terms live in registers after the prologue)
    constexpr int KBYTES = KT * HD * 2, VBYTES = HD * KT * 2;           // 10240 each
    constexpr int STAGE = KBYTES + VBYTES;
    constexpr int NST = 3;                             // stages: tile t + 2 is requested at the top of tile t

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // [stage 0: K | V^T][stage 1][stage 2][constant block 32 B][relcat QT x RCK]; the prologue's kw terms pass through stage 2 (QT x RCK halves)
    T* Cst = reinterpret_cast<T*>(smem_raw + NST * STAGE);
    T* Rc = reinterpret_cast<T*>(smem_raw + NST * STAGE + 32);
    T* Rw = reinterpret_cast<T*>(smem_raw + 2 * STAGE);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int lin = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int qblk = lin % gridDim.x;
    const int q0 = qblk * QT + wave * QW;
    const int sh = lin / gridDim.x;

    const T* __restrict__ Qg = reinterpret_cast<const T*>(p.Q) + (long)sh * p.L * HD;
    const T* __restrict__ Kg = reinterpret_cast<const T*>(p.K) + (long)sh * p.L * HD;
    const T* __restrict__ Vg = reinterpret_cast<const T*>(p.Vt) + (long)sh * HD * p.Lp;

    if (tid < 16) Cst[tid] = (T)((tid < 2) ? 1.f : 0.f);
    const float c1 = p.scale * LOG2E;

    // ---- DMA of key tile kt into stage st: pieces u = wave + 4 t, t = 0 .. 2 (10 of K, 10 of V^T)
    auto dma_k = [&](int kt, int st) {
        const unsigned char* kb = reinterpret_cast<const unsigned char*>(Kg + (long)kt * KT * HD);
        const unsigned kbl = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)kb);
        const unsigned kbh = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)kb >> 32));
        const unsigned char* kbase = reinterpret_cast<const unsigned char*>(((unsigned long long)kbh << 32) | kbl);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int u = wave + 4 * t;
            if (u < 10) {                                   // wave-uniform
                const unsigned koff = (unsigned)(u * 1024 + lane * 16);
                const unsigned kdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)st * STAGE + (unsigned)u * 1024u);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(koff), "s"(kbase), "s"(kdst) : "memory");
            }
        }
    };
    auto dma_v = [&](int kt, int st) {
        const unsigned char* vb = reinterpret_cast<const unsigned char*>(Vg + (long)kt * KT);
        const unsigned vbl = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)vb);
        const unsigned vbh = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)vb >> 32));
        const unsigned char* vbase = reinterpret_cast<const unsigned char*>(((unsigned long long)vbh << 32) | vbl);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int u = wave + 4 * t;
            if (u < 10) {
                const int j = u * 64 + lane;                // LDS slot (row d, physical piece c): source piece c ^ ((d >> 1) & 7)
                const int d = j >> 3, c = (j & 7) ^ ((d >> 1) & 7);
                const unsigned voff = (unsigned)(d * p.Lp * 2 + c * 16);
                const unsigned vdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)st * STAGE + (unsigned)KBYTES + (unsigned)u * 1024u);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(vbase), "s"(vdst) : "memory");
            }
        }
    };
    auto dma_tile = [&](int kt, int st) {
        const unsigned char* kb = reinterpret_cast<const unsigned char*>(Kg + (long)kt * KT * HD);
        const unsigned char* vb = reinterpret_cast<const unsigned char*>(Vg + (long)kt * KT);
        const unsigned kbl = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)kb);
        const unsigned kbh = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)kb >> 32));
        const unsigned vbl = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)vb);
        const unsigned vbh = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)vb >> 32));
        const unsigned char* kbase = reinterpret_cast<const unsigned char*>(((unsigned long long)kbh << 32) | kbl);
        const unsigned char* vbase = reinterpret_cast<const unsigned char*>(((unsigned long long)vbh << 32) | vbl);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int u = wave + 4 * t;
            if (u < 10) {                                   // wave-uniform
                const unsigned koff = (unsigned)(u * 1024 + lane * 16);
                const unsigned kdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)st * STAGE + (unsigned)u * 1024u);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(koff), "s"(kbase), "s"(kdst) : "memory");
                const int j = u * 64 + lane;                // LDS slot (row d, physical piece c): source piece c ^ ((d >> 1) & 7)
                const int d = j >> 3, c = (j & 7) ^ ((d >> 1) & 7);
                const unsigned voff = (unsigned)(d * p.Lp * 2 + c * 16);
                const unsigned vdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)st * STAGE + (unsigned)KBYTES + (unsigned)u * 1024u);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(vbase), "s"(vdst) : "memory");
            }
        }
    };
    const int ntiles = (DBG & 256) ? 1 : p.nk / KT;
    dma_tile(0, 0);
    if (ntiles > 1) dma_tile(1, 1);

    // ---- Q fragments, pre-scaled
    Frag qf[2][NKS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int row = q0 + qb * 16 + li;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d0 = ks * 32 + g * 8;
            qf[qb][ks] = (row < p.L && d0 < HD) ? TR::load_frag(Qg + (long)row * HD + d0) : TR::zero_frag();
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qb][ks].v[j] = (half_t)((float)qf[qb][ks].v[j] * c1);
        }
    }

    // ---- relcat (BIAS 2).  On this path KW == 64 and a wave's 32 queries lie in ONE grid row qy, columns x0 .. x0 + 31 (x0 = 0 | 32), so the table rows a wave
    // needs are known up front — kh terms: rows qy .. qy + KH - 1 (kh = qy - j + KH - 1), kw terms: rows x0 .. x0 + 94 (kw = qx - j + 63) — ten blocks of 16 rows
    // (attn2_kernel walks all sixteen blocks of both tables behind divisions by the runtime KW and tests each), and the fragments of block b + 1 are fetched
    // while block b runs through its MFMAs and stores: measured with the loop cut to one key tile, the prologue was 1.5 ms of the kernel's 7.5
    // (profiles/r05_m_attn2d_ablation.txt).  Every (query, kh < KH) and (query, kw < 64) entry is written, so the rows need no initialisation.
    const float inv_scale = 1.0f / p.scale;
    if (q0 < p.L) {
        T* myrc = Rc + (wave * QW) * RCK;
        T* myrw = Rw + (wave * QW) * RCK;
        const int qy = __builtin_amdgcn_readfirstlane(q0 >> 6), x0 = __builtin_amdgcn_readfirstlane(q0 & 63);
        const int nbh = (p.KH + 15) >> 4, nblk = nbh + 6;
        const bool kh4 = (p.KH & 3) == 0;
        auto load_tab = [&](int blk, Frag (&tf)[NKS]) {
            const bool hh = blk < nbh;
            const float* __restrict__ tab = hh ? p.tab_h : p.tab_w;
            const int nj = hh ? 2 * p.KH - 1 : 2 * KT - 1;
            const int j = (hh ? qy + 16 * blk : x0 + 16 * (blk - nbh)) + li;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int d0 = ks * 32 + g * 8;
                tf[ks] = TR::zero_frag();
                if (j < nj && d0 < HD) {
                    const float* src = tab + (long)j * HD + d0;
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(src), t1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { tf[ks].v[e] = (half_t)t0[e]; tf[ks].v[4 + e] = (half_t)t1[e]; }
                }
            }
        };
        auto run_blk = [&](int blk, const Frag (&tf)[NKS]) {
            const bool hh = blk < nbh;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                f32x4 acc = (f32x4)(0.f);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) TR::mma(tf[ks], qf[qb][ks], acc);
                if (hh) {                                   // accumulator row 4g + r is table row qy + 16 blk + 4g + r: kh = KH - 1 - 16 blk - 4g - r
                    const int kh0 = p.KH - 1 - 16 * blk - 4 * g;
                    T* dst = myrc + (qb * 16 + li) * RCK;
                    if (kh4 && kh0 >= 3) {                  // four consecutive kh, descending with r: one 8-byte store
                        Pack4<T>::type v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[3 - r] = (half_t)(acc[r] * inv_scale);
                        *reinterpret_cast<Pack4<T>::type*>(dst + kh0 - 3) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kh0 - r >= 0) dst[kh0 - r] = (half_t)(acc[r] * inv_scale);
                    }
                } else {                                    // table row x0 + 16 b + 4g + r against column x0 + 16 qb + li: kw = 16 (qb - b) + li - 4g - r + 63
                    const int kw0 = 16 * (qb - (blk - nbh)) + li - 4 * g + KT - 1;
                    T* dst = myrw + (qb * 16 + li) * RCK;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kw0 - r >= 0 && kw0 - r < KT) dst[kw0 - r] = (half_t)(acc[r] * inv_scale);
                }
            }
        };
        Frag tfa[NKS], tfb[NKS];
        load_tab(0, tfa);
#pragma unroll 1
        for (int blk = 0; blk < nblk; blk += 2) {
            if (blk + 1 < nblk) load_tab(blk + 1, tfb);
            run_blk(blk, tfa);
            if (blk + 2 < nblk) load_tab(blk + 2, tfa);
            if (blk + 1 < nblk) run_blk(blk + 1, tfb);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // tile 0, the constant block and this wave's relcat rows are in place

    float bw[2][4][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) bw[qb][kb][r] = (float)Rw[(wave * QW + qb * 16 + li) * RCK + kb * 16 + g * 4 + r];
    __syncthreads();                                    // the kw terms are in registers: stage 2 may take its first tile
    if constexpr (PIPE) { if (ntiles > 2) dma_k(2, 2); }

    f32x4 o[2][ND];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int n = 0; n < ND; ++n) o[qb][n] = (f32x4)(0.f);
    float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
    const bool wave_active = q0 < p.L;

    // fragment addresses (bytes, relative to a stage): K rows li of key block kb at kb * 16 * 160; lanes g >= 2 of the last k-step read the constant block
    const unsigned krow = (unsigned)(li * HD * 2);
    const bool pad_lane = g >= 2;
    const unsigned k2_base = pad_lane ? (unsigned)(NST * STAGE) + (unsigned)(g - 2) * 16u : krow + (unsigned)(64 + g * 8) * 2u;
    const unsigned k2_kb = pad_lane ? 0u : (unsigned)(16 * HD * 2);
    const unsigned k2_st = pad_lane ? 0u : (unsigned)STAGE;
    // V^T rows n * 16 + li: physical piece = logical ^ ((row >> 1) & 7); logical piece of (m, half h) = 4 m + 2 h + (g >> 1), byte (g & 1) * 8 inside
    unsigned vsw[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n) vsw[n] = (unsigned)(((n * 16 + li) >> 1) & 7);

    if constexpr (PIPE) {
        constexpr float LAZY_TAU = 8.0f;
        constexpr int PFD = PIPE == 2 ? 2 : 1;              // LDS fragments are read this many steps ahead of their MFMAs
        // S^T of key tile ktn (its K image in stage stn) into sn: the shift rides in the spare contraction slots, the kw terms are the initial value
        auto score = [&](int ktn, int stn, f32x4 (&sn)[2][4]) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float bh = (float)Rc[(wave * QW + qb * 16 + li) * RCK + ktn];      // kh == key tile
                const float shift = bh - m_run[qb];
                const half_t hi = (half_t)shift;
                const half_t lo = (half_t)(shift - (float)hi);
                qf[qb][NKS - 1].v[0] = g == 2 ? hi : qf[qb][NKS - 1].v[0];
                qf[qb][NKS - 1].v[1] = g == 2 ? lo : qf[qb][NKS - 1].v[1];
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sn[qb][kb][r] = bw[qb][kb][r];
            const unsigned char* sK = smem_raw + stn * STAGE;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    Frag kf;
                    if (ks < NKS - 1) kf = TR::load_frag(reinterpret_cast<const T*>(sK + krow + kb * (16 * HD * 2) + (ks * 32 + g * 8) * 2));
                    else kf = TR::load_frag(reinterpret_cast<const T*>(smem_raw + k2_base + (unsigned)stn * k2_st + (unsigned)kb * k2_kb));
                    TR::mma(kf, qf[0][ks], sn[0][kb]);
                    TR::mma(kf, qf[1][ks], sn[1][kb]);
                }
            }
        };
        // the 16-score maximum of one query block's tile (v_max3 chain)
        auto tile_max = [&](const f32x4 (&sx)[4]) {
            float mx = __builtin_fmaxf(__builtin_fmaxf(sx[0][0], sx[0][1]), sx[0][2]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, sx[0][3]), sx[1][0]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, sx[1][1]), sx[1][2]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, sx[1][3]), sx[2][0]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, sx[2][1]), sx[2][2]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, sx[2][3]), sx[3][0]);
            mx = __builtin_fmaxf(__builtin_fmaxf(mx, sx[3][1]), sx[3][2]);
            return __builtin_fmaxf(mx, sx[3][3]);
        };
        float tm[2] = {0.f, 0.f};                               // maxima of the scores in sc (computed at the end of the tile that produced them)
        // One key tile.  [A] the lazy-maximum test of sc (rare: raise the reference), then ONE stretch of 44 MFMAs in 22 steps, each step fenced so that the
        // issue order is the source order: S^T(kt + 1) -> sn in 12 steps (K fragment of the next step prefetched; the shift pair, then the exponentials /
        // row sums / conversions of sc's key blocks 0, 1, 2 ride beside them), PV(kt) part m = 0 in 5 steps (beside them: key block 3), part m = 1 in 5 steps
        // (beside them: the maxima of sn for the next tile's test).
        auto tile = [&](auto last_tag, int kt, int st, f32x4 (&sc)[2][4], f32x4 (&sn)[2][4]) {
            constexpr bool LAST = decltype(last_tag)::value;
            const bool k3 = kt + 3 < ntiles, v2 = kt + 2 < ntiles;                      // block-uniform
            if (!(DBG & 128)) {
                if (k3) dma_k(kt + 3, st);                     // K(kt) was read in tile kt - 1, V(kt - 1) too: every wave is past the barrier that ended it
                if (v2) dma_v(kt + 2, st >= 1 ? st - 1 : 2);
            }
            if (wave_active) {
                if (!(DBG & 64) && (kt == 0 || __any((tm[0] > LAZY_TAU) || (tm[1] > LAZY_TAU)))) {
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        float mx = tm[qb];
                        mx = fmaxf(mx, __shfl_xor(mx, 16));
                        mx = fmaxf(mx, __shfl_xor(mx, 32));
                        const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
                        m_run[qb] += delta;
                        l_run[qb] *= alpha;
#pragma unroll
                        for (int n = 0; n < ND; ++n)
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[qb][n][r] *= alpha;
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) sc[qb][kb][r] -= delta;
                    }
                }
                Frag pf[2][2];
                float rs[2] = {0.f, 0.f};
                if constexpr ((DBG & 64) != 0) {
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                        for (int m = 0; m < 2; ++m) pf[qb][m] = qf[qb][m];
                }
                auto soft4 = [&](int qb, int kb) {             // four scores of sc: exp2, row sum, fp16 P^T slots
                    if constexpr ((DBG & 64) != 0) return;
                    float pv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pv[r] = (DBG & 1) ? sc[qb][kb][r] * 0.5f : __builtin_amdgcn_exp2f(sc[qb][kb][r]);
                        pf[qb][kb >> 1].v[(kb & 1) * 4 + r] = (half_t)pv[r];
                    }
                    // the row sum as one opaque chain: left to itself the compiler pairs the two query blocks' sums into packed adds at the END of the tile and
                    // keeps all 32 exponentials live.  The leading s_nop is the wait state a vector instruction needs behind the transcendental that produced its
                    // operand (the hazard recogniser does not look inside inline assembly).
                    asm("s_nop 0\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4"
                        : "+v"(rs[qb]) : "v"(pv[0]), "v"(pv[1]), "v"(pv[2]), "v"(pv[3]));
                };
                const unsigned char* sV = smem_raw + st * STAGE + KBYTES;
                auto v_frag = [&](int m, int n) {
                    const unsigned rowb = (unsigned)((n * 16 + li) * 128) + (unsigned)(g & 1) * 8u;
                    const unsigned p0 = (unsigned)(4 * m + (g >> 1)) ^ vsw[n], p1 = (unsigned)(4 * m + 2 + (g >> 1)) ^ vsw[n];
                    return frag_from_2x4<T>(reinterpret_cast<const T*>(sV + rowb + p0 * 16u), reinterpret_cast<const T*>(sV + rowb + p1 * 16u));
                };
                Frag vq[10];
                vq[0] = v_frag(0, 0);
                const Frag vf0 = vq[0];
                if constexpr (LAST && PFD == 2) vq[1] = v_frag(0, 1);
                if constexpr (!LAST) {
                    const int stn = st == 2 ? 0 : st + 1;
                    const unsigned char* sK = smem_raw + stn * STAGE;
                    auto k_frag = [&](int ks, int kb) {
                        if (ks < NKS - 1) return TR::load_frag(reinterpret_cast<const T*>(sK + krow + kb * (16 * HD * 2) + (ks * 32 + g * 8) * 2));
                        return TR::load_frag(reinterpret_cast<const T*>(smem_raw + k2_base + (unsigned)stn * k2_st + (unsigned)kb * k2_kb));
                    };
                    Frag kq[12];
#pragma unroll
                    for (int i = 0; i < PFD; ++i) kq[i] = k_frag(i >> 2, i & 3);
                    const Frag kf0 = kq[0];
                    float bh[2];
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) bh[qb] = (float)Rc[(wave * QW + qb * 16 + li) * RCK + kt + 1];      // kh == key tile
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 12; ++j) {
                        const int ks = j >> 2, kb = j & 3;
                        const Frag kf = kq[j];
                        if (j + PFD < 12) kq[j + PFD] = (DBG & (16 | 2)) ? kf0 : k_frag((j + PFD) >> 2, (j + PFD) & 3);
                        if (PFD == 2 && j == 10) vq[1] = v_frag(0, 1);
                        if constexpr ((DBG & 2) != 0) {
                            if (ks == 0) {
#pragma unroll
                                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) sn[qb][kb][r] = bw[qb][kb][r];
                            }
                        } else if (ks == 0) {                   // the accumulators start from the kw terms
#pragma unroll
                            for (int qb = 0; qb < 2; ++qb) {
                                f32x4 c0;
#pragma unroll
                                for (int r = 0; r < 4; ++r) c0[r] = bw[qb][kb][r];
                                TR::mma(kf, qf[qb][0], c0);
                                sn[qb][kb] = c0;
                            }
                        } else {
                            TR::mma(kf, qf[0][ks], sn[0][kb]);
                            TR::mma(kf, qf[1][ks], sn[1][kb]);
                        }
                        if (j == 0 || j == 1) {                 // the shift of tile kt + 1 into the spare contraction slots (read by the steps of ks = 2)
                            const int qb = j;
                            const float shift = bh[qb] - m_run[qb];
                            const half_t hi = (half_t)shift;
                            const half_t lo = (half_t)(shift - (float)hi);
                            qf[qb][NKS - 1].v[0] = g == 2 ? hi : qf[qb][NKS - 1].v[0];
                            qf[qb][NKS - 1].v[1] = g == 2 ? lo : qf[qb][NKS - 1].v[1];
                        }
                        // six of the eight softmax pieces ride here (the S^T steps have the MFMA time for them): key blocks 0, 1 (P^T part m = 0), then key block 2
                        if (j == 2) soft4(0, 0);
                        if (j == 4) soft4(0, 1);
                        if (j == 5) soft4(1, 0);
                        if (j == 7) soft4(1, 1);
                        if (j == 8) soft4(0, 2);
                        if (j == 10) soft4(1, 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    soft4(0, 0); soft4(0, 1); soft4(1, 0); soft4(1, 1); soft4(0, 2); soft4(1, 2);
                }
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const int m = j / 5, n = j - m * 5;
                    const Frag vf = vq[j];
                    if (j + PFD < 10) vq[j + PFD] = (DBG & (32 | 4)) ? vf0 : v_frag((j + PFD) / 5, (j + PFD) % 5);
                    if constexpr (!(DBG & 4)) {
                        TR::mma(vf, pf[0][m], o[0][n]);
                        TR::mma(vf, pf[1][m], o[1][n]);
                    } else if (j == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { o[0][0][r] += (float)pf[0][0].v[r] + (float)pf[0][1].v[r]; o[1][0][r] += (float)pf[1][0].v[r] + (float)pf[1][1].v[r]; }
                    }
                    if (j == 0) soft4(0, 3);                    // key block 3 (the last of P^T part m = 1)
                    if (j == 2) soft4(1, 3);
                    if constexpr (!LAST && !(DBG & 64)) { if (j == 6 || j == 8) tm[(j - 6) >> 1] = tile_max(sn[(j - 6) >> 1]); }
                    if constexpr (!LAST && (DBG & 64) != 0) { if (j == 6) { o[0][0][0] += sn[0][0][0] + sn[0][1][1] + sn[0][2][2] + sn[0][3][3]; o[1][0][0] += sn[1][0][0] + sn[1][1][1] + sn[1][2][2] + sn[1][3][3]; } }
                    __builtin_amdgcn_sched_barrier(0);
                }
                l_run[0] += rs[0];
                l_run[1] += rs[1];
            }
            // K(kt + 2) and V(kt + 1), requested a tile ago, have landed; this tile's requests stay in flight (three pieces each on waves 0 / 1, two on waves 2 / 3)
            if constexpr ((DBG & 8) != 0) {
                if (!v2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
                return;
            }
            if (k3) { if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else if (v2) { if (wave < 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        f32x4 sa[2][4], sb[2][4];
        if (wave_active) { score(0, 0, sa); tm[0] = tile_max(sa[0]); tm[1] = tile_max(sa[1]); }
        __syncthreads();                                    // every wave has read K(0): tile 0 re-stages its stage with K(3)
        int st = 0, kt = 0;
        for (; kt + 2 < ntiles; kt += 2) {                  // two tiles per trip: the score registers alternate, no copies
            tile(std::false_type{}, kt, st, sa, sb);
            st = st == 2 ? 0 : st + 1;
            tile(std::false_type{}, kt + 1, st, sb, sa);
            st = st == 2 ? 0 : st + 1;
        }
        if (kt + 2 == ntiles) {
            tile(std::false_type{}, kt, st, sa, sb);
            st = st == 2 ? 0 : st + 1;
            tile(std::true_type{}, kt + 1, st, sb, sa);
        } else {
            tile(std::true_type{}, kt, st, sa, sb);
        }
    } else {
    int st = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        const bool ahead = kt + 2 < ntiles;                 // block-uniform
        if (ahead) dma_tile(kt + 2, st >= 1 ? st - 1 : 2);  // stage (kt + 2) % 3 was read in tile kt - 1: every wave is past the barrier that ended it
        if (wave_active) {
            const unsigned char* sK = smem_raw + st * STAGE;
            const unsigned char* sV = sK + KBYTES;
            float shift[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float bh = (float)Rc[(wave * QW + qb * 16 + li) * RCK + kt];       // kh == kt
                shift[qb] = bh - m_run[qb];
                const half_t hi = (half_t)shift[qb];
                const half_t lo = (half_t)(shift[qb] - (float)hi);
                qf[qb][NKS - 1].v[0] = g == 2 ? hi : qf[qb][NKS - 1].v[0];
                qf[qb][NKS - 1].v[1] = g == 2 ? lo : qf[qb][NKS - 1].v[1];
            }
            f32x4 s[2][4];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qb][kb][r] = bw[qb][kb][r];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    Frag kf;
                    if (ks < NKS - 1) kf = TR::load_frag(reinterpret_cast<const T*>(sK + krow + kb * (16 * HD * 2) + (ks * 32 + g * 8) * 2));
                    else kf = TR::load_frag(reinterpret_cast<const T*>(smem_raw + k2_base + (unsigned)st * k2_st + (unsigned)kb * k2_kb));
                    TR::mma(kf, qf[0][ks], s[0][kb]);
                    TR::mma(kf, qf[1][ks], s[1][kb]);
                }
            }
            constexpr float LAZY_TAU = 8.0f;
            Frag pf[2][2];
            float tmax[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float mx = __builtin_fmaxf(__builtin_fmaxf(s[qb][0][0], s[qb][0][1]), s[qb][0][2]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][0][3]), s[qb][1][0]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][1][1]), s[qb][1][2]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][1][3]), s[qb][2][0]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][2][1]), s[qb][2][2]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][2][3]), s[qb][3][0]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[qb][3][1]), s[qb][3][2]);
                tmax[qb] = __builtin_fmaxf(mx, s[qb][3][3]);
            }
            if (kt == 0 || __any((tmax[0] > LAZY_TAU) || (tmax[1] > LAZY_TAU))) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    float mx = tmax[qb];
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    m_run[qb] += delta;
                    l_run[qb] *= alpha;
#pragma unroll
                    for (int n = 0; n < ND; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[qb][n][r] *= alpha;
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[qb][kb][r] -= delta;
                }
            }
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float rs = 0.f;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(s[qb][kb][r]);
                        rs += pv;
                        pf[qb][kb >> 1].v[(kb & 1) * 4 + r] = (half_t)pv;
                    }
                l_run[qb] += rs;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int n = 0; n < ND; ++n) {
                    const unsigned rowb = (unsigned)((n * 16 + li) * 128) + (unsigned)(g & 1) * 8u;
                    const unsigned p0 = (unsigned)(4 * m + (g >> 1)) ^ vsw[n], p1 = (unsigned)(4 * m + 2 + (g >> 1)) ^ vsw[n];
                    const Frag vf = frag_from_2x4<T>(reinterpret_cast<const T*>(sV + rowb + p0 * 16u), reinterpret_cast<const T*>(sV + rowb + p1 * 16u));
                    TR::mma(vf, pf[0][m], o[0][n]);
                    TR::mma(vf, pf[1][m], o[1][n]);
                }
            }
        }
        // this wave's pieces of tile kt + 1 (requested a tile ago) have landed; those of tile kt + 2, just requested, stay in flight:
        // six LDS-DMA instructions on waves 0 / 1, four on waves 2 / 3 (LDS-DMA loads retire in order among themselves)
        if (ahead) { if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        st = st == 2 ? 0 : st + 1;
    }

    }

    const int s_idx = sh / p.heads, h = sh - s_idx * p.heads;
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float lsum = l_run[qb];
        lsum += __shfl_xor(lsum, 16);
        lsum += __shfl_xor(lsum, 32);
        const int qg = q0 + qb * 16 + li;
        if (qg >= p.L) continue;
        const long row = (long)s_idx * p.ntok + qg;
        const float inv = 1.0f / lsum;
        if (p.out8) { attn_store_mx8(p, o[qb], inv, row, h, g); continue; }
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            Pack4<T>::type v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[qb][n][r] * inv);
            *reinterpret_cast<Pack4<T>::type*>(out + row * p.D + h * HD + n * 16 + g * 4) = v;
        }
    }
}

int launch_attn2d(const AttnParams& p, hipStream_t stream, int pipe) {
    const size_t lds = 3 * (size_t)(KT * 80 * 2 + 80 * KT * 2) + 32 + (size_t)QT * (64 + 8) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2d_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef CVA_ABLATION
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2d_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2d_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((p.L + QT - 1) / QT, p.S * p.heads);
#ifdef CVA_ABLATION
    static const int dbg = cva_env_int("CVA_ATTN2D_DBG", 0);
    if (pipe == 1 && dbg) {
        bool hit = false;
#define CVA_A2D_DBG(D) if (dbg == D) { hit = true; hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2d_kernel<1, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                       hipLaunchKernelGGL((attn2d_kernel<1, D>), grid, dim3(NT), lds, stream, p); }
        CVA_A2D_DBG(1) CVA_A2D_DBG(2) CVA_A2D_DBG(4) CVA_A2D_DBG(6) CVA_A2D_DBG(8) CVA_A2D_DBG(48) CVA_A2D_DBG(64) CVA_A2D_DBG(112) CVA_A2D_DBG(128) CVA_A2D_DBG(136) CVA_A2D_DBG(70) CVA_A2D_DBG(198) CVA_A2D_DBG(256)
#undef CVA_A2D_DBG
        if (!hit) return (int)hipErrorInvalidValue;
        return (int)hipGetLastError();
    }
#endif
#ifdef CVA_ABLATION     // the in-wave pipelined variants are experiment instantiations: they need scratch (20 - 30 spilled VGPRs outside the loop), see the header
    if (pipe == 2) { hipLaunchKernelGGL(attn2d_kernel<2>, grid, dim3(NT), lds, stream, p); return (int)hipGetLastError(); }
    if (pipe == 1) { hipLaunchKernelGGL(attn2d_kernel<1>, grid, dim3(NT), lds, stream, p); return (int)hipGetLastError(); }
#endif
    (void)pipe;
    hipLaunchKernelGGL(attn2d_kernel<0>, grid, dim3(NT), lds, stream, p);
    return (int)hipGetLastError();
}

template <typename T, int HD, int BIAS, int NBK, int VRM = 0>
int launch_attn2_impl(const AttnParams& p, hipStream_t stream) {
    constexpr int HDP = (HD + 31) / 32 * 32;
    size_t lds = (size_t)(KT * lds_pitch<T>(HDP) + HD * lds_pitch<T>(KT)) * sizeof(T);   // (V^T pitch <= lds_pitch)
    if (BIAS == 1) lds += (size_t)(KT + QT) * lds_pitch<T>(NBK * 32) * sizeof(T);
    if (BIAS == 2) lds += (size_t)QT * (128 + 8) * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_kernel<T, HD, BIAS, NBK, VRM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((p.L + QT - 1) / QT, p.S * p.heads);
    hipLaunchKernelGGL((attn2_kernel<T, HD, BIAS, NBK, VRM>), grid, dim3(NT), lds, stream, p);
    return (int)hipGetLastError();
}

template <typename T, int HD>
int launch_attn2_hd(const AttnParams& p, hipStream_t stream) {
    if (p.v_rm) {      // row-major V: fp16, hd 80 only; measured slower than V^T for this kernel (profiles/r04_d_vrm_kernels.txt), so the
                       // instantiations exist in ablation builds only (CVA_VRM_GLOBAL=1) and production never asks (attn_takes_vrm)
#ifdef CVA_ABLATION
        if constexpr (sizeof(T) == 2 && HD == 80) {
            if (!p.tab_h) return launch_attn2_impl<T, HD, 0, 1, 1>(p, stream);
            if (p.KW == KT && p.KH <= 64 && p.nk == p.KH * p.KW) return launch_attn2_impl<T, HD, 2, 1, 1>(p, stream);
            if (p.KH + p.KW <= 32) return launch_attn2_impl<T, HD, 1, 1, 1>(p, stream);
            if (p.KH + p.KW <= 64) return launch_attn2_impl<T, HD, 1, 2, 1>(p, stream);
        }
#endif
        return (int)hipErrorInvalidValue;
    }
    if (!p.tab_h) return launch_attn2_impl<T, HD, 0, 1>(p, stream);
    if constexpr (sizeof(T) == 2 && HD == 80) {      // production geometry of the SAM-H global blocks: the LDS-DMA kernel (global, unwindowed rows: win == 0)
        static const int dma_on = cva_env_int("CVA_ATTN2D", 1);    // 0: attn2_kernel, 1: LDS-DMA ring; ablation library: 2 = + S^T(t + 1) beside the softmax of tile t, 3 = + deeper fragment prefetch
        if (dma_on && p.KW == KT && p.KH <= 64 && p.nk == p.KH * p.KW && p.nk % KT == 0 && p.L == p.nk && p.win == 0 && (p.Lp % 8) == 0 &&
            (((size_t)p.K | (size_t)p.Vt) & 15) == 0)
            return launch_attn2d(p, stream, dma_on - 1);
    }
    if (p.KW == KT && p.KH <= 64 && p.nk == p.KH * p.KW) return launch_attn2_impl<T, HD, 2, 1>(p, stream);
    if (p.KH + p.KW <= 32) return launch_attn2_impl<T, HD, 1, 1>(p, stream);
    if (p.KH + p.KW <= 64) return launch_attn2_impl<T, HD, 1, 2>(p, stream);
    return -1;   // caller falls back to the table-based v1 path
}

}  // namespace

// returns -1 if this geometry is not covered by the fused kernel
template <typename T>
int launch_attention2(const AttnParams& p, hipStream_t stream) {
    switch (p.hd) {
        case 64: return launch_attn2_hd<T, 64>(p, stream);
        case 80: return launch_attn2_hd<T, 80>(p, stream);
        default: return (int)hipErrorInvalidValue;
    }
}

template int launch_attention2<half_t>(const AttnParams&, hipStream_t);
template int launch_attention2<float>(const AttnParams&, hipStream_t);

}  // namespace cva
